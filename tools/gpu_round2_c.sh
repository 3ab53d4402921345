cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02c; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q -k "rccl or batched_ensemble or multi_input or eager" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log)
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "rc $?" >> $O/bench20.err
timeout 600 python bench.py --gpus 2 --backend socket --all-ranks-device 0 --steps 20 --warmup 5 --members 4 --no-single > $O/bench_2ranks_socket.json 2> $O/bench_2ranks_socket.err; echo "rc $?" >> $O/bench_2ranks_socket.err
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?" >> $O/bench_default.err
tail -3 $O/pytest.log; cat $O/bench20.json $O/bench20.err | cut -c1-3000; cat $O/bench_2ranks_socket.json $O/bench_2ranks_socket.err | cut -c1-1500; cat $O/bench_default.json $O/bench_default.err | cut -c1-3000
