#!/bin/bash
# usage (GPU box): tools/pmc_persist.sh TAG "LATTICE N MEMBERS STEPS"  -- SQ counters of the persistent kernels and of the stage launches
# on the same problem (tools/persist_probe.py runs both arms in one process; the counters are told apart by kernel name)
TAG=$1; ARGS=${2:-"quads 128 1 400"}
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o trace -- python3 tools/persist_probe.py $ARGS > $OUT/probe.txt 2> $OUT/stats.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc_sq1 -o pmc -- python3 tools/persist_probe.py $ARGS > /dev/null 2> $OUT/e3
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 tools/persist_probe.py $ARGS > /dev/null 2> $OUT/e4
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 tools/persist_probe.py $ARGS > /dev/null 2> $OUT/e1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- python3 tools/persist_probe.py $ARGS > /dev/null 2> $OUT/e2
python tools/pmc_report.py $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.json
grep -h "k_fwd\|k_adj" $OUT/stats/*/*kernel_stats.csv $OUT/stats/*kernel_stats.csv 2>/dev/null | cut -c1-220 > $OUT/kernel_stats.txt
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/probe.txt; cat $OUT/kernel_stats.txt; cat $OUT/pmc_summary.json
