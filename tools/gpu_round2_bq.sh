cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bq; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log); grep -E "passed|failed" $O/pytest.log | tail -1
for rep in 1 2 3; do
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single > $O/b.json 2> $O/b.err
python - <<PY
import json
l=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("K=20", "%.3e"%l["value"], l["device_ms"], l["grad_norm"])
PY
done
