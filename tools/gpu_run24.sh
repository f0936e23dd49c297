#!/bin/bash
export TMPDIR=/tmp
( timeout 900 python bench.py --workload cityscapes --steps 10 --warmup 3 --no_cpu_baseline ) > gpurun_out/bench_city.log 2>&1; echo "city rc=$?"
grep '^{"metric"' gpurun_out/bench_city.log | cut -c1-2000; tail -3 gpurun_out/bench_city.log | grep -v '^{"metric"' | cut -c1-300
