#!/bin/bash
export TMPDIR=/tmp
( timeout 900 python bench.py --workload pascal_v3plus --steps 6 --warmup 3 --no_cpu_baseline ) > gpurun_out/bench_v3.log 2>&1; echo "bench rc=$?"
grep '^{"metric"' gpurun_out/bench_v3.log | cut -c1-330; tail -3 gpurun_out/bench_v3.log | grep -v '^{"metric"' | cut -c1-300
