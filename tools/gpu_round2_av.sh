cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02av; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_analytic_kat.py tests/test_gpu_edge_cases.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log); tail -3 $O/pytest.log
for rep in 1 2; do
timeout 400 python bench.py --steps 250 --warmup 250 --no-cpu-baseline --no-single > $O/b.json 2> $O/b.err
python - <<PY
import json
l=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("sincos", "%.3e"%l["value"], "adj us", l["roofline"].get("launch_us"), "fwd us", l.get("roofline_forward_kernel",{}).get("launch_us"), l.get("grad_norm"))
PY
done
