cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log)
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
tail -15 $O/pytest.log; cut -c1-400 $O/bench20.json
