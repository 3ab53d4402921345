#!/bin/bash
# usage: tools/ms_sweep.sh "M:S M:S ..." [steps] -- whole-job throughput against (members per GPU, concurrent streams)
for ms in $1; do
  m=${ms%%:*}; s=${ms##*:}
  timeout 300 python bench.py --steps ${2:-2000} --warmup 200 --members $m --streams $s --no-cpu-baseline --no-single 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline_concurrent',{}); print('M=$m S=$s value %.3e fwd_only %.3e concurrent frac fwd %.3f adj %.3f'%(d['value'],d['forward_only_value'],r.get('fwd_frac',-1), r.get('adj_frac',-1)))" || echo "M=$m S=$s failed"
done
