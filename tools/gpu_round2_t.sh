cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/profile_round.sh r02_v3 16 > gpurun_out/prof_r02_v3.log 2>&1
timeout 900 python bench.py --steps 50000 --warmup 250 --no-cpu-baseline --no-single > gpurun_out/prof_r02_v3/bench_50000.json 2> gpurun_out/prof_r02_v3/bench_50000.err
tail -3 gpurun_out/prof_r02_v3.log
