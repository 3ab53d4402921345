cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02w; mkdir -p $O
for CK in auto records stages state segments; do
  echo "== 256 members x 3 inputs (24x16, 4000 steps), sequential engines, DFX_CHECKPOINT=$CK" >> $O/c5_levels.log
  if [ $CK = auto ]; then timeout 300 python tools/stall_probe.py 256 100 0 >> $O/c5_levels.log 2>&1; else DFX_CHECKPOINT=$CK timeout 300 python tools/stall_probe.py 256 100 0 >> $O/c5_levels.log 2>&1; fi
done
cat $O/c5_levels.log
