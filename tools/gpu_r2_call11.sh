#!/bin/bash
# round 2, call 11: epilogue restructured (scale / bias staged in LDS, specialised nests): conv tests, trace, sweep
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q > gpurun_out/r2k_pytest_conv.log 2>&1; echo "conv tests rc=$?"; tail -3 gpurun_out/r2k_pytest_conv.log
timeout 300 python tools/conv_trace.py > gpurun_out/r2k_conv_trace.log 2>&1; echo "trace rc=$?"
grep -E "^==|prologue|per K step \(mean" gpurun_out/r2k_conv_trace.log
export CMS_VARIANTS=0:0,256:0
timeout 300 python tools/conv_variants.py > gpurun_out/r2k_conv_variants.log 2>&1; echo "variants rc=$?"
cat gpurun_out/r2k_conv_variants.log
