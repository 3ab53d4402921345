cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02k; mkdir -p $O
timeout 900 python examples/multi_input_ensemble.py --members 256 --iterations 4 2>&1 | grep -E "designs x 3 inputs|device time" | cut -c1-300 > $O/c5_timing.txt
timeout 900 python -m cProfile -s tottime examples/multi_input_ensemble.py --members 256 --iterations 4 > $O/c5_profile2.txt 2>&1; timeout 900 python examples/multi_input_ensemble.py --members 96 --iterations 4 2>&1 | grep -E "designs x 3 inputs|device time" | cut -c1-300 >> $O/c5_timing.txt
cat $O/c5_timing.txt; grep -n "ncalls" -A16 $O/c5_profile2.txt | cut -c1-150
