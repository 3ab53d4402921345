#!/bin/bash
# usage: MS="16:2 4:4" tools/lib_sweep.sh lib1.so lib2.so ...  -- runs tools/ms_sweep.sh with each library variant copied over libdfx.so
cp difflexmm_amd/libdfx.so /tmp/libdfx_orig.so
for lib in "$@"; do
  cp "$lib" difflexmm_amd/libdfx.so
  echo "== $lib"
  tools/ms_sweep.sh "${MS:-16:2 4:4}" ${STEPS:-2000}
done
cp /tmp/libdfx_orig.so difflexmm_amd/libdfx.so
