#!/bin/bash
# usage: tools/member_sweep.sh [streams] ["M ..."] -- per-launch stage-kernel duration against members per launch (one stream: what rocprof can verify)
S=${1:-1}
for m in ${2:-1 2 4 8 16}; do
  timeout 300 python bench.py --steps 250 --warmup 50 --members $m --streams $S --no-cpu-baseline --no-single --no-as-written 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); a=d['roofline']; r=d['roofline_forward_kernel']; print('streams=$S M=$m value %.3e fwd_only %.3e fwd launch %.2f us (%.3f us/member) frac %.3f adj launch %.2f us (%.3f us/member) frac %.3f'%(d['value'],d['forward_only_value'],r['launch_us'],r['launch_us']/$m,r['frac'],a['launch_us'],a['launch_us']/$m,a['frac']))"
done
