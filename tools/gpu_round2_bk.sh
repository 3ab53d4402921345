cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bk; mkdir -p $O
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.txt 2>&1; tail -5 $O/smoke.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.txt 2>&1; tail -4 $O/bench_driver.txt | cut -c1-400
