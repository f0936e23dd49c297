#!/bin/bash
# round 2, GPU call 7: DeepLab v3+ without library convolutions / BatchNorm: tests, bench, rocprof kernel list
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_deeplab3plus.py tests/test_gpu_conv.py tests/test_gpu_bn.py -q -m gpu > gpurun_out/r2g_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2g_tests.log
timeout 300 python bench.py --workload pascal_v3plus --no_cpu_baseline --steps 20 > gpurun_out/r2g_bench_v3.log 2> gpurun_out/r2g_bench_v3.err; echo "rc=$?" >> gpurun_out/r2g_bench_v3.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2g_prof -o r2g -- python $GRAFT_REPO_ROOT/bench.py --workload pascal_v3plus --steps 5 --warmup 2 --no_cpu_baseline --no_roofline_events > $GRAFT_REPO_ROOT/gpurun_out/r2g_prof.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/r2g_prof/r2g_results.db 60 > gpurun_out/r2g_kernel_stats_v3plus.csv 2>> gpurun_out/r2g_prof.log; rm -rf gpurun_out/r2g_prof
grep -E "passed|failed|rc=|^FAILED" gpurun_out/r2g_tests.log | tail -n 8; tail -c 700 gpurun_out/r2g_bench_v3.log; tail -n 3 gpurun_out/r2g_bench_v3.err; cut -c1-110 gpurun_out/r2g_kernel_stats_v3plus.csv | head -n 40
