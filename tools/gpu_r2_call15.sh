#!/bin/bash
# round 2, call 15: buffer-addressed direct-to-LDS loader (variant 40) vs default: bitwise check + timing, trace
mkdir -p gpurun_out
export CMS_VARIANTS=0:0,0:40,64:0,64:40
timeout 300 python tools/conv_variants.py > gpurun_out/r2o_conv_variants.log 2>&1; echo "variants rc=$?"
cat gpurun_out/r2o_conv_variants.log
unset CMS_VARIANTS
CMS_TRACE_VARIANT=41 timeout 300 python tools/conv_trace.py > gpurun_out/r2o_conv_trace_bufa.log 2>&1; echo "trace rc=$?"
grep -E "^==|prologue|per K step \(mean" gpurun_out/r2o_conv_trace_bufa.log
