cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02an; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log); tail -3 $O/pytest.log
for m in 16; do
timeout 400 python bench.py --steps 250 --warmup 250 --members $m --no-cpu-baseline > $O/b_$m.json 2> $O/b_$m.err
python - <<PY
import json
l=json.loads(open("$O/b_$m.json").read().strip().splitlines()[-1])
print("members $m", "%.3e"%l["value"], "adj us", l["roofline"].get("launch_us"), l["roofline"]["frac"], "fwd us", l.get("roofline_forward_kernel",{}).get("launch_us"), l.get("grad_norm"))
PY
done
