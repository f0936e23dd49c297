#!/bin/bash
# round 2, call 10: cycle trace of the default convolution kernel
mkdir -p gpurun_out
timeout 300 python tools/conv_trace.py > gpurun_out/r2j_conv_trace.log 2>&1; echo "trace rc=$?"
cat gpurun_out/r2j_conv_trace.log
