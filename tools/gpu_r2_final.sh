#!/bin/bash
# round 2, last call: smoke, the whole -m gpu suite and the default bench line on the final tree
mkdir -p gpurun_out
( time timeout 600 python __graft_entry__.py smoke ) > gpurun_out/r2fin_smoke.log 2>&1; echo "smoke rc=$?" | tee gpurun_out/r2fin_summary.log
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r2fin_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2fin_summary.log
( time timeout 600 python bench.py ) > gpurun_out/r2fin_bench.log 2> gpurun_out/r2fin_bench.err; echo "bench rc=$?" | tee -a gpurun_out/r2fin_summary.log
tail -n 3 gpurun_out/r2fin_smoke.log
grep -E "^FAILED|passed|failed" gpurun_out/r2fin_pytest_gpu.log | tail -n 6
grep '^{"metric"' gpurun_out/r2fin_bench.log | cut -c1-260
