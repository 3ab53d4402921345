#!/bin/bash
# usage: tools/member_sweep.sh [streams] -- per-launch stage-kernel duration against members per launch (one stream: what rocprof can verify)
S=${1:-1}
for m in 1 2 4 8 16; do
  timeout 300 python bench.py --steps 1000 --warmup 100 --members $m --streams $S --no-cpu-baseline --no-single 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d['roofline_adjoint_kernel']; print('streams=$S M=$m value %.3e fwd_only %.3e fwd launch %.2f us frac %.3f adj launch %.2f us frac %.3f'%(d['value'],d['forward_only_value'],r['launch_us'],r['frac'],a['launch_us'],a['frac']))"
done
