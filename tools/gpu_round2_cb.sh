cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02cb; mkdir -p $O
for v in base npb base npb; do
  if [ $v = base ]; then unset DFX_LIBRARY; else export DFX_LIBRARY=$GRAFT_REPO_ROOT/difflexmm_amd/libdfx_$v.so; fi
  timeout 600 python tools/c4_problem_timing.py 32 4000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c40-200 | sed "s/^/$v: /"
done
for v in base npb; do
  if [ $v = base ]; then unset DFX_LIBRARY; else export DFX_LIBRARY=$GRAFT_REPO_ROOT/difflexmm_amd/libdfx_$v.so; fi
  timeout 400 python bench.py --steps 250 --warmup 250 --no-cpu-baseline --no-single > $O/b.json 2> $O/b.err
  python - <<PY
import json
l=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("$v quads", "%.3e"%l["value"], "adj us", l["roofline"].get("launch_us"), "fwd us", l.get("roofline_forward_kernel",{}).get("launch_us"), l.get("grad_norm"))
PY
done
