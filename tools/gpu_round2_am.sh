cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02am; mkdir -p $O
for v in base abl1; do
  if [ $v = base ]; then unset DFX_LIBRARY; else export DFX_LIBRARY=$GRAFT_REPO_ROOT/difflexmm_amd/libdfx_$v.so; fi
  for m in 1 16; do
  timeout 400 python bench.py --steps 250 --warmup 250 --members $m --streams 1 --no-cpu-baseline --no-single > $O/b_${v}_$m.json 2> $O/b_${v}_$m.err
  python - <<PY
import json
l=json.loads(open("$O/b_${v}_$m.json").read().strip().splitlines()[-1])
print("$v members $m", "%.3e"%l["value"], "adj us", l["roofline"].get("launch_us"), "fwd us", l.get("roofline_forward_kernel",{}).get("launch_us"), l.get("grad_norm"))
PY
  done
done
