cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02m; mkdir -p $O
cp difflexmm_amd/libdfx.so /tmp/libdfx_orig.so
for v in base adj5 adj6 fwd6 fwd4; do
  cp variants/$v.so difflexmm_amd/libdfx.so
  for S in 1 2; do
    echo "== $v DFX_STREAMS=$S" >> $O/occ.log
    DFX_STREAMS=$S timeout 300 python tools/k20_probe.py 250 16 2 >> $O/occ.log 2>&1
  done
done
cp /tmp/libdfx_orig.so difflexmm_amd/libdfx.so
cat $O/occ.log
