#!/bin/bash
# usage (GPU box): tools/scratch/ab_libs.sh OUTFILE lib1 lib2 ...   ("default" = the product library)
OUT=$1; shift
A1="bench.py --streams 1 --members 16 --steps 250 --warmup 0 --no-cpu-baseline --no-single --no-roofline-leg --no-as-written --no-launch-bound"
A2="bench.py --steps 250 --warmup 250 --no-cpu-baseline --no-single --no-roofline-leg --no-as-written --no-launch-bound"
for rep in 1 2 3; do
for lib in "$@"; do
  if [ "$lib" = default ]; then unset DFX_LIBRARY; else export DFX_LIBRARY=$PWD/variants/libdfx_$lib.so; fi
  for mode in 1 2; do
    if [ $mode = 1 ]; then A=$A1; else A=$A2; fi
    timeout 300 python $A 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
dm=d['device_ms']
print('$lib', 'streams', d['config']['concurrent_streams'], 'value %.4e'%d['value'], 'fwd %.2f adj %.2f ms'%(dm['forward'],dm['adjoint']), 'level', d['config']['checkpoint'])
" >> $OUT
  done
done
done
cat $OUT
