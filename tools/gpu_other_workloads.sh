#!/bin/bash
# The two other BASELINE configurations with the current code (bench lines only).
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python bench.py --workload cityscapes --steps 10 --warmup 3 --no_cpu_baseline ) > gpurun_out/bench_city.log 2>&1; echo "city rc=$?"
grep '^{"metric"' gpurun_out/bench_city.log | cut -c1-220
( timeout 900 python bench.py --workload pascal_v3plus --steps 6 --warmup 3 --no_cpu_baseline ) > gpurun_out/bench_v3.log 2>&1; echo "v3 rc=$?"
grep '^{"metric"' gpurun_out/bench_v3.log | cut -c1-220
