#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_executor.py -m gpu -q -p no:cacheprovider --timeout 600 ) > gpurun_out/pytest_conv.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/summary.log
grep -E "^FAILED|passed|failed|^E  " gpurun_out/pytest_conv.log | head -20
( timeout 600 python tools/conv_bench.py 20 ) > gpurun_out/conv_bench.log 2>&1
tail -22 gpurun_out/conv_bench.log
( timeout 900 python bench.py --steps 10 --warmup 3 --no_cpu_baseline ) > gpurun_out/bench_hip.log 2>&1
grep '^{"metric"' gpurun_out/bench_hip.log | cut -c1-260
