cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bv; mkdir -p $O
for rep in 1 2; do (timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_$rep.log 2>&1; echo "pytest rc $?" >> $O/pytest_$rep.log); grep -E "passed|failed|rc " $O/pytest_$rep.log | tail -2; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
