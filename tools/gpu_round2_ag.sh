cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ag; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -q -x -k "adaptive" > $O/pytest_adaptive.log 2>&1; echo "rc $?" >> $O/pytest_adaptive.log)
tail -3 $O/pytest_adaptive.log
timeout 600 python tools/adaptive_probe.py 16 2 1e-8 > $O/adaptive_probe_after.txt 2>&1
cat $O/adaptive_probe_after.txt
