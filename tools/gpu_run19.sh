#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && ( timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no_cpu_baseline --no_roofline_events ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
python tools/rocpd_timeline.py gpurun_out/prof_tmp/bench_results.db | tail -4
python tools/rocpd_timeline.py gpurun_out/prof_tmp/bench_results.db gaps 2>&1 | tee gpurun_out/gaps.log | head -60
rm -rf gpurun_out/prof_tmp
