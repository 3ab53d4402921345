cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02br; mkdir -p $O
true
tail -4 $O/pytest.log
bash tools/profile_round.sh r02_v9 16 > $O/profile_round.log 2>&1
tail -30 $O/profile_round.log
