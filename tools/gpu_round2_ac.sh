cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ac; mkdir -p $O
timeout 600 python tools/merged_inputs_probe.py 96 > $O/merged_probe_96_fixed.txt 2>&1
for m in 96 256; do
timeout 900 python examples/multi_input_ensemble.py --members $m --iterations 4 2>&1 | grep -E "designs x 3 inputs|device time" | cut -c1-420 >> $O/c5_after.txt
done
cat $O/merged_probe_96_fixed.txt $O/c5_after.txt
