cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02y; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log)
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
timeout 400 python bench.py --steps 250 --warmup 250 --no-cpu-baseline > $O/bench250.json 2> $O/bench250.err
timeout 400 python bench.py --steps 250 --warmup 250 --no-cpu-baseline --contact-cutoff-deg 60 --contact-min-deg 30 > $O/bench250_contact.json 2> $O/bench250_contact.err
timeout 900 python examples/multi_input_ensemble.py --members 256 --iterations 4 2>&1 | grep -E "designs x 3 inputs|device time" | cut -c1-300 > $O/c5_timing.txt
tail -3 $O/pytest.log; cat $O/c5_timing.txt
for f in bench20 bench250 bench250_contact; do python - <<PY
import json
l=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
print("$f", "%.3e"%l["value"], l["roofline"]["kernel"] if "kernel" in l["roofline"] else "", l["roofline"].get("launch_us"), l["roofline"]["frac"], l.get("roofline_forward_kernel",{}).get("launch_us"), l.get("objective"), l.get("grad_norm"))
PY
done
