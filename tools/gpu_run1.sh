#!/bin/bash
# first GPU pass: smoke, parity tests, conv-library probe, short bench, rocprof kernel stats
mkdir -p gpurun_out
export TMPDIR=/tmp
export MIOPEN_FIND_MODE=${MIOPEN_FIND_MODE:-FAST}
echo "=== rocm-smi" > gpurun_out/env.log; rocm-smi --showproductname >> gpurun_out/env.log 2>&1; nproc >> gpurun_out/env.log
( time timeout 900 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/summary.log
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 -x --deselect tests/test_gpu_parity.py::test_ema_full_size_bit_exact 2>&1 | tail -80 ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=${PIPESTATUS[0]}" | tee -a gpurun_out/summary.log
tail -30 gpurun_out/smoke.log
tail -60 gpurun_out/pytest_gpu.log
