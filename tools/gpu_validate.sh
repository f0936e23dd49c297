#!/bin/bash
# Round-end validation on a GPU box: smoke, the whole -m gpu suite, the default bench line, a rocprofv3 kernel trace of
# the bench (summaries land in gpurun_out/; copy what should be kept into profiles/).
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee gpurun_out/summary.log
( time timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/summary.log
( time timeout 900 python bench.py ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/summary.log
cd /tmp && ( timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_round -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no_cpu_baseline ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?" | tee -a $GRAFT_REPO_ROOT/gpurun_out/summary.log
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof_round/bench_results.db 60 > gpurun_out/kernel_stats_round.csv 2>&1
python tools/rocpd_timeline.py gpurun_out/prof_round/bench_results.db > gpurun_out/timeline_round.csv 2>&1
rm -rf gpurun_out/prof_round
tail -3 gpurun_out/smoke.log
grep -E "^FAILED|passed|failed" gpurun_out/pytest_gpu.log | tail -12
grep '^{"metric"' gpurun_out/bench.log | cut -c1-400
grep '^{"metric"' gpurun_out/rocprof.log | cut -c1-200
head -8 gpurun_out/kernel_stats_round.csv | cut -c1-160
tail -5 gpurun_out/timeline_round.csv
