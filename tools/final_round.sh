#!/bin/bash
# usage (on the GPU box): tools/final_round.sh TAG   -- everything DESIGN.md section 4 quotes for a round, into gpurun_out/final_TAG/
TAG=${1:-rXX}; OUT=gpurun_out/final_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/gpu_pytest.txt
tools/profile_round.sh $TAG 16 > $OUT/profile_round.log 2>&1
# (the whole 50 000-step C3 as written is the `c3_as_written` leg of the default bench line since round 4: profile_round.sh's bench.json)
timeout 900 python bench.py --workload c5 > $OUT/bench_c5_256_designs.json 2> $OUT/bench_c5.err
timeout 600 python bench.py --workload c4 --steps 4000 --warmup 1 > $OUT/bench_c4_64_designs.json 2> $OUT/bench_c4.err
timeout 600 python bench.py --workload c2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout 600 python bench.py --workload paper > $OUT/bench_paper.json 2> $OUT/bench_paper.err      # the notebooks' own call: adaptive Dopri5 + gradient, 24 x 16
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_paper -o trace -- python3 bench.py --workload paper --no-cpu-baseline > $OUT/bench_paper_under_rocprof.json 2> $OUT/stats_paper.err
find $OUT/stats_paper -name "*_kernel_trace.csv" -size +2M -delete; find $OUT/stats_paper -name "*.db" -delete
timeout 900 python bench.py --workload c4 --steps 75000 > $OUT/bench_c4_64_designs_full_horizon.json 2> $OUT/bench_c4_full.err      # config 4 as written: 75 000 steps (two engine calls of 32 designs on one GPU)
timeout 600 python bench.py --workload c5 --c5-members 32 --c5-iterations 4 --no-cpu-baseline > $OUT/bench_c5_32_designs.json 2> $OUT/bench_c5_32.err
for a in "quads 128 1 2000 1 1 DFX_PERSIST=1,DFX_CHECKPOINT=segments DFX_PERSIST=1,DFX_CHECKPOINT=segments,DFX_SEG_OVERLAP=0" "quads 128 1 400" "quads 128 2 400" "quads 128 4 400" "kagome 64 8 400" "quads 64 8 400" "quads 32 1 2000 0 0" "quads 32 64 1000 0 0"; do timeout 300 python tools/persist_probe.py $a; done > $OUT/persist_probe.txt 2>/dev/null
timeout 600 python tools/c4_problem_timing.py 8 4000 > $OUT/c4_8_designs.txt 2>&1
timeout 900 python examples/multi_input_ensemble.py --members 256 --iterations 4 2>&1 | grep -E "designs x|evaluations of|device time" > $OUT/c5_256_designs_4_iterations.txt
timeout 600 python bench.py --gpus 2 --backend socket --all-ranks-device 0 --steps 250 --warmup 250 --members 8 --no-cpu-baseline > $OUT/bench_two_rank_rehearsal_socket.json 2> $OUT/bench_two_rank.err
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
tail -3 $OUT/gpu_pytest.txt; cat $OUT/c5_256_designs_4_iterations.txt; tail -2 $OUT/smoke.txt; tail -c 600 $OUT/bench_c5_256_designs.json
