cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02x; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log)
timeout 900 python examples/multi_input_ensemble.py --members 256 --iterations 4 2>&1 | grep -E "designs x 3 inputs|device time" | cut -c1-300 > $O/c5_timing.txt
timeout 900 python examples/multi_input_ensemble.py --members 96 --iterations 4 2>&1 | grep -E "designs x 3 inputs|device time" | cut -c1-300 >> $O/c5_timing.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
tail -3 $O/pytest.log; cat $O/c5_timing.txt; cut -c1-300 $O/bench20.json
