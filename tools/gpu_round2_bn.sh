cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bn; mkdir -p $O
for m in 32 96; do for ck in stages records; do
DFX_CHECKPOINT=$ck timeout 900 python examples/multi_input_ensemble.py --members $m --iterations 4 2>&1 | grep -E "designs x 3 inputs|device time" | cut -c1-130 | sed "s/^/$ck: /" >> $O/c5.txt
done; done
cat $O/c5.txt
