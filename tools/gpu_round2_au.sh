cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02au; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "design_subset" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log); tail -15 $O/pytest.log
