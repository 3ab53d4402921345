cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02p; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log)
for CK in segments state stages; do
  echo "== DFX_CHECKPOINT=$CK K=2000" >> $O/probe.log
  DFX_CHECKPOINT=$CK timeout 300 python tools/k20_probe.py 2000 16 2 >> $O/probe.log 2>&1
done
timeout 900 python bench.py --steps 50000 --warmup 250 --no-cpu-baseline --no-single > $O/bench_50000.json 2> $O/bench_50000.err; echo "rc $?" >> $O/bench_50000.err
tail -4 $O/pytest.log; cat $O/probe.log; cut -c1-900 $O/bench_50000.json; tail -3 $O/bench_50000.err
