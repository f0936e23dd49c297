export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/r04bd_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload pascal --steps 12 --warmup 6 --no_cpu_baseline --traffic omit --no_also > $OUT/r04bd_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $OUT/r04bd_prof/bench_results.db 40 > $OUT/r04bd_kernel_stats.csv 2>>$OUT/r04bd_rocprof.log
python tools/step_timeline.py $OUT/r04bd_prof/bench_results.db 10 > $OUT/r04bd_timeline.txt 2>>$OUT/r04bd_rocprof.log
rm -rf $OUT/r04bd_prof
grep '^{"metric"' $OUT/r04bd_rocprof.log | cut -c1-150
head -12 $OUT/r04bd_kernel_stats.csv | cut -c1-150
