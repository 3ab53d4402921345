#!/bin/bash
# usage (GPU box): tools/pmc_quick.sh TAG [env assignments...]  -- kernel stats + SQ / traffic counters of the two stage kernels, one stream, 16 members
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --streams 1 --members 16 --steps 100 --warmup 0 --no-cpu-baseline --no-single --no-roofline-leg --no-as-written"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o trace -- python3 $ARGS > /dev/null 2> $OUT/stats.err
export DFX_DUAL_CHAIN=0
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > /dev/null 2> $OUT/e1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > /dev/null 2> $OUT/e2
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc_sq1 -o pmc -- python3 $ARGS > /dev/null 2> $OUT/e3
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $ARGS > /dev/null 2> $OUT/e4
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq3 -o pmc -- python3 $ARGS > /dev/null 2> $OUT/e5
python tools/pmc_report.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3 > $OUT/pmc_summary.json
grep -h "k_fwd\|k_adj" $OUT/stats/*/*kernel_stats.csv $OUT/stats/*kernel_stats.csv 2>/dev/null | cut -c1-200 > $OUT/kernel_stats.txt
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/kernel_stats.txt; cat $OUT/pmc_summary.json
