#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_executor.py -m gpu -q -x -p no:cacheprovider --timeout 300 ) > gpurun_out/pytest_exec.log 2>&1; echo "pytest rc=$?"
grep -E "^E  |passed|failed|Error" gpurun_out/pytest_exec.log | cut -c1-250 | head -20
( timeout 300 python tools/epi_probe.py ) 2>&1 | grep -v amdgpu.ids
( timeout 600 python bench.py --no_cpu_baseline ) > gpurun_out/bench_tmp.log 2>&1; echo "bench rc=$?"
grep '^{"metric"' gpurun_out/bench_tmp.log; tail -3 gpurun_out/bench_tmp.log | grep -v '^{"metric"' | cut -c1-300
