cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bz; mkdir -p $O
timeout 600 python tools/c5_call_probe.py 256 > $O/probe.txt 2>&1
DFX_EAGER_STEPS=0 timeout 600 python tools/c5_call_probe.py 256 >> $O/probe.txt 2>&1
grep "M =" $O/probe.txt
