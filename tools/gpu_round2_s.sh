cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02s; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log)
for S in 1 2; do echo "== spill fix, DFX_STREAMS=$S K=250" >> $O/probe.log; DFX_STREAMS=$S timeout 300 python tools/k20_probe.py 250 16 2 >> $O/probe.log 2>&1; done
DFX_CHECKPOINT=segments timeout 300 python tools/k20_probe.py 2000 16 2 >> $O/probe.log 2>&1
tail -4 $O/pytest.log; cat $O/probe.log
