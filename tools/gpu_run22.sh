#!/bin/bash
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "tile or epilogue" ) > gpurun_out/pytest_exec.log 2>&1; echo "pytest rc=$?"
grep -E "^E  |passed|failed|Error" gpurun_out/pytest_exec.log | cut -c1-250 | head -10
timeout 300 python tools/variant_probe.py 0,5 2>&1 | grep -v amdgpu.ids
