#!/bin/bash
# round 2, GPU call 3: HIP stem + single-pass ASPP head: tests, full suite, bench, rocprof kernel stats
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_stem.py tests/test_gpu_programs.py "tests/test_gpu_hip_engine_parity.py" -q -m gpu -x -s > gpurun_out/r2c_new.log 2>&1; echo "rc=$?" >> gpurun_out/r2c_new.log
timeout 600 python -m pytest tests -q -m gpu --deselect tests/test_gpu_stem.py --deselect tests/test_gpu_programs.py --deselect tests/test_gpu_hip_engine_parity.py > gpurun_out/r2c_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r2c_pytest_gpu.log
timeout 400 python bench.py --no_cpu_baseline > gpurun_out/r2c_bench.log 2> gpurun_out/r2c_bench.err; echo "rc=$?" >> gpurun_out/r2c_bench.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2c_prof -o r2c -- python $GRAFT_REPO_ROOT/bench.py --workload pascal --steps 10 --warmup 3 --no_cpu_baseline --no_roofline_events > $GRAFT_REPO_ROOT/gpurun_out/r2c_prof.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/r2c_prof > gpurun_out/r2c_kernel_stats.csv 2>> gpurun_out/r2c_prof.log || true
grep -E "passed|failed|rc=" gpurun_out/r2c_new.log | tail -n 3; grep -E "passed|failed|rc=" gpurun_out/r2c_pytest_gpu.log | tail -n 3; tail -c 400 gpurun_out/r2c_bench.log; tail -n 3 gpurun_out/r2c_bench.err; head -n 12 gpurun_out/r2c_kernel_stats.csv
