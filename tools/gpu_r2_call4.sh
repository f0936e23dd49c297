#!/bin/bash
# round 2, GPU call 4: stem / programs / parity / BN / N1 mIoU tests, then the rest of the suite
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stem.py tests/test_gpu_bn.py tests/test_gpu_programs.py tests/test_gpu_miou_training.py tests/test_gpu_hip_engine_parity.py -q -m gpu -s > gpurun_out/r2d_new.log 2>&1; echo "rc=$?" >> gpurun_out/r2d_new.log
timeout 600 python -m pytest tests -q -m gpu --deselect tests/test_gpu_stem.py --deselect tests/test_gpu_bn.py --deselect tests/test_gpu_programs.py --deselect tests/test_gpu_miou_training.py --deselect tests/test_gpu_hip_engine_parity.py > gpurun_out/r2d_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r2d_pytest_gpu.log
grep -E "passed|failed|rc=|^FAILED|N1 mIoU" gpurun_out/r2d_new.log | tail -n 12; grep -E "passed|failed|rc=|^FAILED" gpurun_out/r2d_pytest_gpu.log | tail -n 6
