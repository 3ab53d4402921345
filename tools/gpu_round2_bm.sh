cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bm; mkdir -p $O
for cfg in "16 2" "16 4" "18 3" "24 2" "24 3" "32 2" "32 4"; do set -- $cfg
timeout 400 python bench.py --steps 250 --warmup 250 --members $1 --streams $2 --no-cpu-baseline --no-single --no-roofline-leg > $O/b.json 2> $O/b.err
python - <<PY
import json
l=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
rc=l.get("roofline_concurrent") or {}
print("members $1 streams $2", "%.3e"%l["value"], l["config"]["checkpoint"], rc.get("fwd_stage_period_us"), rc.get("adj_stage_period_us"))
PY
done
