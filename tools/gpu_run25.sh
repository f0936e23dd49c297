#!/bin/bash
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "wgrad" ) > gpurun_out/pytest_exec.log 2>&1; echo "pytest rc=$?"
grep -E "^E  |passed|failed|Error" gpurun_out/pytest_exec.log | cut -c1-250 | head -10
timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids
( timeout 600 python bench.py --no_cpu_baseline --no_roofline_events ) > gpurun_out/bench_tmp.log 2>&1; echo "bench rc=$?"
grep '^{"metric"' gpurun_out/bench_tmp.log | cut -c1-200
