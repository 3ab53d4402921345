cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ae; mkdir -p $O
for v in ilp memclause relaxed base; do
  if [ $v = base ]; then unset DFX_LIBRARY; else export DFX_LIBRARY=$GRAFT_REPO_ROOT/difflexmm_amd/libdfx_$v.so; fi
  timeout 400 python bench.py --steps 250 --warmup 250 --no-cpu-baseline > $O/bench250_$v.json 2> $O/bench250_$v.err
  python - <<PY
import json
l=json.loads(open("$O/bench250_$v.json").read().strip().splitlines()[-1])
print("$v", "%.3e"%l["value"], l["roofline"].get("launch_us"), l["roofline"]["frac"], l.get("roofline_forward_kernel",{}).get("launch_us"), l.get("grad_norm"))
PY
done
