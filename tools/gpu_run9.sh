#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && ( timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no_cpu_baseline ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof_tmp/bench_results.db 40 > gpurun_out/kernel_stats_tmp.csv 2>&1
rm -rf gpurun_out/prof_tmp
head -16 gpurun_out/kernel_stats_tmp.csv | cut -c1-150
( timeout 900 python bench.py --workload cityscapes --steps 6 --warmup 2 --no_cpu_baseline ) > gpurun_out/bench_city.log 2>&1; echo "city rc=$?"
grep '^{"metric"' gpurun_out/bench_city.log | cut -c1-330; tail -3 gpurun_out/bench_city.log | cut -c1-300
