cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ax; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log)
tail -4 $O/pytest.log
bash tools/profile_round.sh r02_v6 16 > $O/profile_round.log 2>&1
tail -30 $O/profile_round.log
