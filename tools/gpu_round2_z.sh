cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02z; mkdir -p $O
for w in 0 8 16 32; do
timeout 900 python examples/multi_input_ensemble.py --members 256 --iterations 4 --host-workers $w 2>&1 | grep -E "designs x 3 inputs|device time" | cut -c1-420 >> $O/c5_workers.txt
done
timeout 900 python tools/c5_host_profile.py --members 256 --iterations 4 --host-workers 16 2>&1 | grep -A 40 "function calls" | cut -c1-200 > $O/c5_profile_workers.txt
cat $O/c5_workers.txt; head -50 $O/c5_profile_workers.txt
