cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bd; mkdir -p $O
timeout 900 python bench.py --steps 50000 --warmup 250 --no-cpu-baseline --no-single > $O/bench_full.json 2> $O/bench_full.err
python - <<PY
import json
l=json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
print("full", "%.3e"%l["value"], l["config"]["checkpoint"], l["config"]["members_per_gpu"], l["device_ms"], l["end_to_end_frac_of_hbm_peak"], l["objective"][:2], l["grad_norm"])
PY
