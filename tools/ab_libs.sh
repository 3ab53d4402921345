#!/bin/bash
# usage: tools/ab_libs.sh libA.so libB.so ... -- A/B of engine builds on one box: per-launch leg (1 stream) and the timed job (2 streams), each
# library twice, interleaved (DFX_LIBRARY selects the build; every one must be a gfx950 engine)
for rep in 1 2; do
  for lib in "$@"; do
    DFX_LIBRARY=$PWD/$lib timeout 300 python bench.py --steps ${STEPS:-250} --warmup 50 --no-cpu-baseline --no-as-written 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); a=d['roofline']; r=d['roofline_forward_kernel']; s=d.get('single_system',{}); print('$lib rep $rep value %.4e fwd_only %.4e fwd launch %.2f us adj launch %.2f us single %.3e'%(d['value'],d['forward_only_value'],r['launch_us'],a['launch_us'],s.get('value',0)))"
  done
done
