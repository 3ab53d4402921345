cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bu; mkdir -p $O
timeout 300 tools/mock/fused_pair_mock 16 > $O/mock.txt 2>&1
timeout 300 tools/mock/fused_pair_mock 4 >> $O/mock.txt 2>&1
timeout 300 tools/mock/fused_pair_mock 1 >> $O/mock.txt 2>&1
cat $O/mock.txt
