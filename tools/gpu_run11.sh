#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_executor.py -m gpu -q -x -p no:cacheprovider --timeout 300 ) > gpurun_out/pytest_exec.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_exec.log
for flags in "--no_overlap --no_roofline_events" "--no_roofline_events" ""; do
  ( timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline $flags ) > gpurun_out/bench_tmp.log 2>&1
  echo "flags=[$flags] rc=$?"; grep '^{"metric"' gpurun_out/bench_tmp.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', 'enqueue', round(d['config']['host_enqueue_ms_per_step'],2), 'roof', round(d['roofline']['achieved'],1), d['roofline']['avg_launch_ms'])
"
  tail -2 gpurun_out/bench_tmp.log | cut -c1-200 | grep -v '^{"metric"' | grep -v amdgpu.ids
done
( timeout 300 python tools/tail_probe.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tail_probe.log
( timeout 300 python tools/conv_bench.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/conv_bench.log
