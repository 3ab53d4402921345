cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bt; mkdir -p $O
for v in base sb base sb; do
  if [ $v = base ]; then unset DFX_LIBRARY; else export DFX_LIBRARY=$GRAFT_REPO_ROOT/difflexmm_amd/libdfx_$v.so; fi
  timeout 400 python bench.py --steps 250 --warmup 250 --no-cpu-baseline --no-single > $O/b_${v}.json 2> $O/b_${v}.err
  python - <<PY
import json
l=json.loads(open("$O/b_${v}.json").read().strip().splitlines()[-1])
print("$v", "%.3e"%l["value"], "adj us", l["roofline"].get("launch_us"), "fwd us", l.get("roofline_forward_kernel",{}).get("launch_us"), l.get("grad_norm"))
PY
done
