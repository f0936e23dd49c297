#!/bin/bash
# round 2, GPU call 1: fp32 parity engine + HIP-engine parity vs the oracle, full GPU suite, bench with both shapes
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout 600 python -m pytest tests/test_gpu_hip_engine_parity.py -q -m gpu -s > gpurun_out/r2_parity.log 2>&1; echo "rc=$?" >> gpurun_out/r2_parity.log
timeout 600 python -m pytest tests -q -m gpu --deselect tests/test_gpu_hip_engine_parity.py > gpurun_out/r2_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r2_pytest_gpu.log
timeout 400 python bench.py > gpurun_out/r2_bench.log 2> gpurun_out/r2_bench.err; echo "rc=$?" >> gpurun_out/r2_bench.err
tail -5 gpurun_out/r2_parity.log gpurun_out/r2_pytest_gpu.log; tail -c 1500 gpurun_out/r2_bench.log
