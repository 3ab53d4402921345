#!/bin/bash
# usage: tools/launch_bound_ab.sh libA.so libB.so ... -- launch-bound regime (1 and 4 members per launch, 1 stream): forward and reverse launch
# period of each engine build, three interleaved repetitions
for rep in 1 2 3; do
  for lib in "$@"; do
    for m in 1 4; do
      DFX_LIBRARY=$PWD/$lib timeout 300 python bench.py --members $m --steps 500 --warmup 50 --streams 1 --no-cpu-baseline --no-single --no-as-written --no-roofline-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); dm=d['device_ms']; print('$lib rep $rep members $m: fwd %.3f us/launch adj %.3f us/launch value %.4e'%(1e3*dm['forward']/3000, 1e3*dm['adjoint']/3000, d['value']))"
    done
  done
done
