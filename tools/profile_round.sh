#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh TAG [MEMBERS]
# Collects what DESIGN.md section 4 cites, into gpurun_out/prof_TAG/:
#   bench.json            default bench line
#   stats/                rocprofv3 --kernel-trace --stats of `bench.py --streams 1` (one stream: the regime the profiler can observe)
#   pmc_fetch|pmc_write|pmc_sq*/   PMC passes (separate runs, --kernel-trace only), DFX_DUAL_CHAIN=0
TAG=${1:-rXX}; M=${2:-16}
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
ARGS="bench.py --streams 1 --members $M --steps 250 --warmup 0 --no-cpu-baseline --no-single --no-roofline-leg --no-as-written --no-launch-bound"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o trace -- python3 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
PARGS="bench.py --streams 1 --members $M --steps 250 --warmup 0 --no-cpu-baseline --no-single --no-roofline-leg --no-as-written --no-launch-bound"
export DFX_DUAL_CHAIN=0
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $PARGS > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- python3 $PARGS > /dev/null 2> $OUT/pmc_write.err
timeout 300 rocprofv3 --pmc TCC_HIT TCC_MISS --kernel-trace --output-format csv -d $OUT/pmc_tcc -o pmc -- python3 $PARGS > /dev/null 2> $OUT/pmc_tcc.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc_sq1 -o pmc -- python3 $PARGS > /dev/null 2> $OUT/pmc_sq1.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $PARGS > /dev/null 2> $OUT/pmc_sq2.err
python tools/pmc_report.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_tcc $OUT/pmc_sq1 $OUT/pmc_sq2 > $OUT/pmc_summary.json
# keep only the summaries (the raw traces are large)
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*_counter_collection.csv" -size +2M -delete
find $OUT -name "*.db" -delete
ls -la $OUT $OUT/stats 2>/dev/null | head -40
cat $OUT/pmc_summary.json | head -60
