cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log)
for cfg in "128 2" "0 2" "128 1" "0 1"; do set -- $cfg
  echo "== DFX_EAGER_STEPS=$1 DFX_STREAMS=$2" >> $O/k20.log
  DFX_EAGER_STEPS=$1 DFX_STREAMS=$2 timeout 300 python tools/k20_probe.py 20 16 5 >> $O/k20.log 2>&1
done
echo "== K=100 eager vs graph" >> $O/k20.log
DFX_EAGER_STEPS=128 timeout 300 python tools/k20_probe.py 100 16 3 >> $O/k20.log 2>&1
DFX_EAGER_STEPS=0 timeout 300 python tools/k20_probe.py 100 16 3 >> $O/k20.log 2>&1
timeout 300 python tools/k20_probe.py 5000 16 2 >> $O/k20.log 2>&1
tail -4 $O/pytest.log; cat $O/k20.log
