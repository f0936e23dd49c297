#!/bin/bash
export TMPDIR=/tmp
( time timeout 900 python bench.py --workload pascal_v3plus --steps 6 --warmup 3 --no_cpu_baseline ) > gpurun_out/bench_v3.log 2>&1; echo "bench rc=$?"
grep '^{"metric"' gpurun_out/bench_v3.log | cut -c1-260; grep real gpurun_out/bench_v3.log
( time timeout 900 python bench.py --steps 10 --warmup 3 --no_cpu_baseline ) > gpurun_out/bench_tmp.log 2>&1; echo "bench rc=$?"
grep '^{"metric"' gpurun_out/bench_tmp.log | cut -c1-260; grep real gpurun_out/bench_tmp.log
