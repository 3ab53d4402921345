cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
nproc > $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>&1; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())" >> $O/host.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20_nocpu.json 2> $O/bench20_nocpu.err; echo "rc $?" >> $O/bench20_nocpu.err
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "rc $?" >> $O/bench20.err
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?" >> $O/bench_default.err
cat $O/host.txt; for f in bench20_nocpu bench20 bench_default; do echo "== $f"; cut -c1-1200 $O/$f.json; tail -3 $O/$f.err; done
