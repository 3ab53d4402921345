#!/bin/bash
# usage (GPU box): tools/persist_ab.sh OUTFILE name1 name2 ...  -- A/B of builds of the persistent kernels ("default" or variants/libdfx_NAME.so): device time
# per stage of one 128x128 system, 8 kagome designs and 4 quads systems, three interleaved repetitions
OUT=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset DFX_LIBRARY; else export DFX_LIBRARY=$PWD/variants/libdfx_$lib.so; fi
    for a in "quads 128 1 800" "kagome 64 8 400" "quads 128 4 400"; do
      timeout 300 python tools/persist_probe.py $a 2>/dev/null | grep "'DFX_PERSIST': '1'" | sed "s/^/$lib: /" >> $OUT
    done
  done
done
cat $OUT
