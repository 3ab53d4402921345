cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bi; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "group_streams" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log); tail -12 $O/pytest.log
