cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ap; mkdir -p $O
for a in 0 2 1 3; do
  export DFX_ABLATE=$a
  timeout 400 python bench.py --steps 250 --warmup 250 --forward-only --streams 1 --no-cpu-baseline --no-single > $O/b_$a.json 2> $O/b_$a.err
  python - <<PY
import json
try:
    l=json.loads(open("$O/b_$a.json").read().strip().splitlines()[-1])
    print("ablate $a", "%.3e"%l["value"], "fwd us", l["roofline"].get("launch_us"))
except Exception as e:
    print("ablate $a failed", open("$O/b_$a.err").read()[-300:])
PY
done
