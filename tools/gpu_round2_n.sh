cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02n; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log)
for S in 1 2; do
  echo "== flags-in-dictionary-byte DFX_STREAMS=$S" >> $O/probe.log
  DFX_STREAMS=$S timeout 300 python tools/k20_probe.py 250 16 2 >> $O/probe.log 2>&1
done
DFX_STREAMS=2 timeout 300 python tools/k20_probe.py 20 16 4 >> $O/probe.log 2>&1
tail -4 $O/pytest.log; cat $O/probe.log
