cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bj; mkdir -p $O
for ck in records stages segments; do DFX_CHECKPOINT=$ck timeout 600 python tools/fd_check_fullsize.py 1000 1e-5 >> $O/fd.txt 2>&1; done
cat $O/fd.txt | grep -v amdgpu.ids
