cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ay; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log); tail -3 $O/pytest.log
for ck in records records stages; do
  DFX_CHECKPOINT=$ck timeout 400 python bench.py --steps 250 --warmup 250 --no-cpu-baseline --no-single > $O/b_$ck.json 2> $O/b_$ck.err
  python - <<PY
import json
l=json.loads(open("$O/b_$ck.json").read().strip().splitlines()[-1])
print("$ck", "%.3e"%l["value"], l["config"]["checkpoint"], "adj us", l["roofline"].get("launch_us"), "fwd us", l.get("roofline_forward_kernel",{}).get("launch_us"), l.get("grad_norm"))
PY
done
