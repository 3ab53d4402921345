cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bs; mkdir -p $O
timeout 900 python bench.py --steps 50000 --warmup 250 --no-cpu-baseline --no-single --input-delay 3.3333333e-3 --target-shift 21 25 > $O/bench_full_c3_text.json 2> $O/err.txt
python - <<PY
import json
l=json.loads(open("$O/bench_full_c3_text.json").read().strip().splitlines()[-1])
print("full C3 as in the text", "%.3e"%l["value"], l["config"]["checkpoint"], l["config"]["input_delay_s"], l["config"]["target_blocks"], l["device_ms"], l["objective"][:3], l["grad_norm"])
PY
