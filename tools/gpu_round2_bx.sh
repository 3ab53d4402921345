cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bx; mkdir -p $O
timeout 900 python tools/c5_host_profile.py --members 256 --iterations 4 2>&1 | grep -v "^round\|^ [0-9 .\]]*$\|^\[" | grep -A 34 "function calls" | head -80 | cut -c1-180 > $O/c5_profile.txt
cat $O/c5_profile.txt | head -45
