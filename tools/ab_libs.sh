#!/bin/bash
# usage (GPU box): tools/ab_libs.sh OUTFILE name1 name2 ...   -- A/B of engine builds on one box, three repetitions, interleaved:
# "default" = the product library, NAME = variants/libdfx_NAME.so (through DFX_LIBRARY); per build the one-stream leg and the
# two-stream job: value, device time of the forward pass and of the reverse sweep
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
