#!/bin/bash
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -p no:cacheprovider --timeout 600 -k "wgrad" ) > gpurun_out/pytest_exec.log 2>&1; echo "pytest conv rc=$?"
grep -E "^E  |passed|failed|Error" gpurun_out/pytest_exec.log | cut -c1-250 | head -12
( timeout 900 python -m pytest tests/test_gpu_deeplab3plus.py -m gpu -q -x -p no:cacheprovider --timeout 600 ) > gpurun_out/pytest_v3.log 2>&1; echo "pytest v3 rc=$?"
grep -E "^E  |passed|failed|Error" gpurun_out/pytest_v3.log | cut -c1-250 | head -30
