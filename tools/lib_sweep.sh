#!/bin/bash
# usage: tools/lib_sweep.sh lib1.so lib2.so ...  -- runs the bench (no cpu baseline) with each library variant copied over libdfx.so
cp difflexmm_amd/libdfx.so /tmp/libdfx_orig.so
for lib in "$@"; do
  cp "$lib" difflexmm_amd/libdfx.so
  for m in 1 4; do
    python bench.py --steps 2500 --members $m --no-cpu-baseline --no-single 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib M=$m value %.3e fwd_only %.3e roofline launch %.2f us adj %.2f us'%(d['value'],d['forward_only_value'],d['roofline']['launch_us'],d['roofline_adjoint_kernel']['launch_us']))"
  done
done
cp /tmp/libdfx_orig.so difflexmm_amd/libdfx.so
