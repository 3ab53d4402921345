cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log)
tail -25 $O/pytest.log; cat $O/smoke.log | tail -3
