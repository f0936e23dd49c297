#!/bin/bash
# round 2, call 42: kernel list of the DeepLab v3+ step (cfg 4) after the host fix
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2al_prof -o v3 -- python $GRAFT_REPO_ROOT/bench.py --workload pascal_v3plus --steps 6 --warmup 2 --no_cpu_baseline --no_roofline_events > $GRAFT_REPO_ROOT/gpurun_out/r2al_prof.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/r2al_prof/v3_results.db 45 > gpurun_out/r2al_kernel_stats_v3plus.csv 2>> gpurun_out/r2al_prof.log; rm -rf gpurun_out/r2al_prof
cut -c1-150 gpurun_out/r2al_kernel_stats_v3plus.csv | head -36
