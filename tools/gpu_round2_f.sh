cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log)
for CK in records stages; do for S in 2 1; do
  echo "== DFX_CHECKPOINT=$CK DFX_STREAMS=$S" >> $O/probe.log
  DFX_CHECKPOINT=$CK DFX_STREAMS=$S timeout 300 python tools/k20_probe.py 20 16 3 >> $O/probe.log 2>&1
  DFX_CHECKPOINT=$CK DFX_STREAMS=$S timeout 300 python tools/k20_probe.py 1000 16 2 >> $O/probe.log 2>&1
done; done
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err
tail -5 $O/pytest.log; cat $O/probe.log; cut -c1-600 $O/bench20.json
