#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -p no:cacheprovider --timeout 300 -k "wgrad" ) > gpurun_out/pytest_conv.log 2>&1; echo "pytest conv rc=$?" | tee gpurun_out/summary.log
grep -E "^FAILED|passed|failed|^E  " gpurun_out/pytest_conv.log | head -40
