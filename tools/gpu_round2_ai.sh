cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ai; mkdir -p $O; rm -f $O/c4.txt
(timeout 1200 python -m pytest tests/test_gpu_problems.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log); tail -3 $O/pytest.log
for b in 8 32; do timeout 600 python tools/c4_timing.py $b 4000 2>&1 | grep -v amdgpu.ids >> $O/c4.txt; done
cat $O/c4.txt
