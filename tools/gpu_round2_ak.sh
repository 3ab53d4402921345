cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ak; mkdir -p $O
export DFX_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c4 -- python3 tools/c4_problem_timing.py 32 400 > $O/c4_under_rocprof.txt 2>&1
head -6 $O/stats/c4_kernel_stats.csv | cut -c1-260
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o pmc -- python3 tools/c4_problem_timing.py 32 400 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o pmc -- python3 tools/c4_problem_timing.py 32 400 > /dev/null 2>&1
python tools/pmc_report.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.json; cat $O/pmc_summary.json | head -20
find $O -name "*_kernel_trace.csv" -size +2M -delete; find $O -name "*_counter_collection.csv" -size +2M -delete
tail -2 $O/c4_under_rocprof.txt | cut -c1-300
