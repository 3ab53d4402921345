cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bf; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log); grep -E "passed|failed" $O/pytest.log | tail -1
for ck in stages records; do
DFX_CHECKPOINT=$ck timeout 900 python examples/multi_input_ensemble.py --members 256 --iterations 4 2>&1 | grep -E "designs x 3 inputs|device time" | cut -c1-200 | sed "s/^/$ck: /" >> $O/c5.txt
done
cat $O/c5.txt
timeout 400 python bench.py --steps 250 --warmup 250 --no-cpu-baseline --no-single > $O/b.json 2> $O/b.err
python - <<PY
import json
l=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("bench250", "%.3e"%l["value"], l["config"]["checkpoint"], "adj us", l["roofline"].get("launch_us"), "fwd us", l.get("roofline_forward_kernel",{}).get("launch_us"), l.get("grad_norm"))
PY
