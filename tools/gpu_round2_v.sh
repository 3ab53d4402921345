cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02v; mkdir -p $O
for M in 8 12 16 20 24; do
  echo "== members=$M DFX_STREAMS=2 K=250" >> $O/m.log
  DFX_STREAMS=2 timeout 300 python tools/k20_probe.py 250 $M 2 >> $O/m.log 2>&1
done
cat $O/m.log
