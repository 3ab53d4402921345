cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ca; mkdir -p $O
for b in 8 32; do timeout 600 python tools/c4_problem_timing.py $b 4000 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/c4p.txt; done
for m in 96 256; do timeout 900 python examples/multi_input_ensemble.py --members $m --iterations 4 2>&1 | grep -E "designs x 3 inputs|device time|evaluations of the whole" | cut -c1-200 >> $O/c5.txt; done
cat $O/c4p.txt $O/c5.txt
