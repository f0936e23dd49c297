#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -p no:cacheprovider --timeout 300 ) > gpurun_out/pytest_conv.log 2>&1; echo "pytest conv rc=$?" | tee gpurun_out/summary.log
( timeout 600 python tools/conv_bench.py 20 ) > gpurun_out/conv_bench.log 2>&1; echo "conv bench rc=$?" | tee -a gpurun_out/summary.log
( timeout 60 ./tools/tr_probe.bin ) > gpurun_out/tr_probe.log 2>&1; echo "probe rc=$?" | tee -a gpurun_out/summary.log
grep -E "^FAILED|passed|failed|^E  " gpurun_out/pytest_conv.log | head -40
cat gpurun_out/conv_bench.log | tail -25
