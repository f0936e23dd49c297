#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_executor.py -m gpu -q -p no:cacheprovider --timeout 600 -s ) > gpurun_out/pytest_exec.log 2>&1; echo "pytest exec rc=$?" | tee gpurun_out/summary.log
grep -E "^FAILED|passed|failed|^E  |Error" gpurun_out/pytest_exec.log | head -40
( timeout 900 python bench.py --steps 10 --warmup 3 --no_cpu_baseline ) > gpurun_out/bench_hip.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/summary.log
tail -4 gpurun_out/bench_hip.log | cut -c1-900
