cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
(timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02a/pytest.log)
timeout 300 python tools/k20_probe.py 20 16 6 > gpurun_out/r02a/k20.log 2>&1
timeout 300 python tools/k20_probe.py 5000 16 2 > gpurun_out/r02a/k5000.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02a/trace -o k20 -- python3 tools/k20_probe.py 20 16 3 > gpurun_out/r02a/k20_traced.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench20.json 2> gpurun_out/r02a/bench20.err
tail -5 gpurun_out/r02a/pytest.log; cat gpurun_out/r02a/k20.log gpurun_out/r02a/k5000.log; ls -la gpurun_out/r02a/trace/* | head
