cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02e; mkdir -p $O
for K in 250 1000 5000; do for E in 0 1000000; do for S in 2 1; do
  echo "== K=$K DFX_EAGER_STEPS=$E DFX_STREAMS=$S" >> $O/eager.log
  DFX_EAGER_STEPS=$E DFX_STREAMS=$S timeout 300 python tools/k20_probe.py $K 16 2 >> $O/eager.log 2>&1
done; done; done
cat $O/eager.log
