cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02r; mkdir -p $O
for rep in 1 2; do for E in 0 1000000; do for S in 1 2; do
  echo "== rep $rep DFX_EAGER_STEPS=$E DFX_STREAMS=$S K=250" >> $O/ab.log
  DFX_EAGER_STEPS=$E DFX_STREAMS=$S timeout 300 python tools/k20_probe.py 250 16 2 >> $O/ab.log 2>&1
done; done; done
cat $O/ab.log
