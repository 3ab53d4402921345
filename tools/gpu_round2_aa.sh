cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02aa; mkdir -p $O
for v in nopark park3 park4; do
  export DFX_LIBRARY=$GRAFT_REPO_ROOT/difflexmm_amd/libdfx_$v.so
  (timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_$v.log 2>&1; echo "pytest rc $?" >> $O/pytest_$v.log)
  timeout 400 python bench.py --steps 250 --warmup 250 --no-cpu-baseline > $O/bench250_$v.json 2> $O/bench250_$v.err
  tail -2 $O/pytest_$v.log
  python - <<PY
import json
l=json.loads(open("$O/bench250_$v.json").read().strip().splitlines()[-1])
print("$v", "%.3e"%l["value"], l["roofline"].get("launch_us"), l["roofline"]["frac"], l.get("roofline_forward_kernel",{}).get("launch_us"), l.get("grad_norm"))
PY
done
