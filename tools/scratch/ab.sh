#!/bin/bash
for i in 1 2 3; do timeout 300 python bench.py --steps 250 --warmup 5 --no-single --no-cpu-baseline --as-written-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.4g adj %.2f fwd %.2f  conc fwd %.2f adj %.2f  dev %s gn %.12g' % (d['value'], d['roofline']['launch_us'], d['roofline_forward_kernel']['launch_us'], d['roofline_concurrent']['fwd_stage_period_us'], d['roofline_concurrent']['adj_stage_period_us'], d['device_ms'], d['grad_norm']))
"; done
