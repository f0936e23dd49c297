#!/bin/bash
# usage: gpu_quick.sh <pytest args...>   -- a focused GPU test run, log in gpurun_out/pytest_quick.log
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest "$@" -m gpu -q -x -p no:cacheprovider --timeout 600 ) > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"
grep -E "^E  |passed|failed|Error" gpurun_out/pytest_quick.log | cut -c1-250 | head -30
