#!/bin/bash
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_deeplab3plus.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout 600 -k "deeplab3plus or iteration or step" ) > gpurun_out/pytest_v3.log 2>&1; echo "pytest rc=$?"
grep -E "^E  |passed|failed|Error" gpurun_out/pytest_v3.log | cut -c1-250 | head -30
( time timeout 900 python bench.py --workload pascal_v3plus --steps 6 --warmup 3 --no_cpu_baseline ) > gpurun_out/bench_v3.log 2>&1; echo "bench rc=$?"
grep '^{"metric"' gpurun_out/bench_v3.log | cut -c1-260
