cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02l; mkdir -p $O
for M in 256 96 32; do for E in 0 1000000; do
  echo "== members=$M DFX_EAGER_STEPS=$E" >> $O/c5_eager_vs_graph.txt
  DFX_EAGER_STEPS=$E timeout 600 python examples/multi_input_ensemble.py --members $M --iterations 3 2>&1 | grep -E "designs x 3 inputs|device time" | cut -c1-220 >> $O/c5_eager_vs_graph.txt
done; done
cat $O/c5_eager_vs_graph.txt
