cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02by; mkdir -p $O
for e in default 0; do for m in 32 256; do
  if [ $e = default ]; then unset DFX_EAGER_STEPS; else export DFX_EAGER_STEPS=$e; fi
  timeout 900 python examples/multi_input_ensemble.py --members $m --iterations 4 2>&1 | grep -E "designs x 3 inputs|device time" | cut -c1-140 | sed "s/^/eager_steps=$e: /" >> $O/c5.txt
done; done
cat $O/c5.txt
