#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for flags in "--no_roofline_events" ""; do
  ( timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline $flags ) > gpurun_out/bench_tmp.log 2>&1
  echo "flags=[$flags] rc=$?"; grep '^{"metric"' gpurun_out/bench_tmp.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', 'enqueue', round(d['config']['host_enqueue_ms_per_step'],2), 'roof', round(d['roofline']['achieved'],1), d['roofline']['avg_launch_ms'])
"
done
cd /tmp && ( timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no_cpu_baseline --no_roofline_events ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof_tmp/bench_results.db 40 > gpurun_out/kernel_stats_tmp.csv 2>&1
python tools/rocpd_timeline.py gpurun_out/prof_tmp/bench_results.db 2>&1 | tee gpurun_out/timeline.log
rm -rf gpurun_out/prof_tmp
head -12 gpurun_out/kernel_stats_tmp.csv | cut -c1-150
