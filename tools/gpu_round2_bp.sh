cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bp; mkdir -p $O
for s in 1 2 4; do DFX_STREAMS=$s timeout 600 python tools/c4_problem_timing.py 8 4000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-260 | sed "s/^/streams $s: /" >> $O/c4.txt; done
cat $O/c4.txt
