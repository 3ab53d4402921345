cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02u; mkdir -p $O
cp difflexmm_amd/libdfx.so /tmp/libdfx_orig.so
for rep in 1 2; do for v in base nt; do
  cp variants/$v.so difflexmm_amd/libdfx.so
  for S in 1 2; do
    echo "== rep $rep $v DFX_STREAMS=$S" >> $O/nt.log
    DFX_STREAMS=$S timeout 300 python tools/k20_probe.py 250 16 2 >> $O/nt.log 2>&1
  done
done; done
cp /tmp/libdfx_orig.so difflexmm_amd/libdfx.so
cat $O/nt.log
