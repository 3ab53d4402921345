cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bw; mkdir -p $O
for d in 0 1; do for m in 1 4 16; do
  if [ $d = 1 ]; then export DFX_NO_DICT=1; else unset DFX_NO_DICT; fi
  timeout 400 python bench.py --steps 250 --warmup 250 --members $m --streams 1 --no-cpu-baseline --no-single > $O/b.json 2> $O/b.err
  python - <<PY
import json
l=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("nodict $d members $m", "%.3e"%l["value"], "adj us", l["roofline"].get("launch_us"), "fwd us", l.get("roofline_forward_kernel",{}).get("launch_us"), l.get("grad_norm"))
PY
done; done
