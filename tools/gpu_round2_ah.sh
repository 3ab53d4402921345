cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ah; mkdir -p $O
for b in 8 32; do timeout 600 python tools/c4_timing.py $b 4000 >> $O/c4.txt 2>&1; done
DFX_STREAMS=1 timeout 600 python tools/c4_timing.py 32 1000 >> $O/c4.txt 2>&1
timeout 900 python examples/multi_input_ensemble.py --members 256 --iterations 4 2>&1 | grep -E "designs x 3 inputs|device time" | cut -c1-300 >> $O/c5.txt
cat $O/c4.txt $O/c5.txt
