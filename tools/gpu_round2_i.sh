cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
for M in 16 32 64; do for S in 2 4; do for K in 20 250; do
  echo "== M=$M DFX_STREAMS=$S K=$K" >> $O/ms.log
  DFX_STREAMS=$S timeout 300 python tools/k20_probe.py $K $M 3 >> $O/ms.log 2>&1
done; done; done
cat $O/ms.log
