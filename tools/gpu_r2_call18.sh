#!/bin/bash
# round 2, call 18: direct-to-LDS weight-gradient kernel: tests, A/B against the register loader, split targets, trace
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_hip_engine_parity.py -m gpu -x -q > gpurun_out/r2r_pytest.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2r_pytest.log
for cfg in "0 384" "1 384" "1 512" "1 768" "1 1024"; do
  set -- $cfg
  echo "== CMS_WGRAD_DMA=$1 CMS_WGRAD_TARGET=$2"
  CMS_WGRAD_DMA=$1 CMS_WGRAD_TARGET=$2 timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r2r_wgrad_bench.log 2>&1
cat gpurun_out/r2r_wgrad_bench.log
timeout 200 python tools/wgrad_trace.py l3 l2 > gpurun_out/r2r_wgrad_trace.log 2>&1; cat gpurun_out/r2r_wgrad_trace.log
