#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee gpurun_out/summary.log
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | tail -120 ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a gpurun_out/summary.log
( time timeout 900 python bench.py --steps 5 --warmup 2 ) > gpurun_out/bench_short.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/summary.log
tail -15 gpurun_out/smoke.log
tail -70 gpurun_out/pytest_gpu.log
tail -5 gpurun_out/bench_short.log
