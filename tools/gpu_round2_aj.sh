cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02aj; mkdir -p $O
for b in 8 32; do timeout 600 python tools/c4_problem_timing.py $b 4000 2>&1 | grep -v amdgpu.ids >> $O/c4p.txt; done
cat $O/c4p.txt
