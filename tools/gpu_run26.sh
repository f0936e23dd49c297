#!/bin/bash
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -q -x -p no:cacheprovider --timeout 600 ) > gpurun_out/pytest_dist.log 2>&1; echo "pytest rc=$?"
grep -E "^E  |passed|failed|Error" gpurun_out/pytest_dist.log | cut -c1-250 | head -30
