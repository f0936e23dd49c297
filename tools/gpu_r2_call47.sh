#!/bin/bash
# round 2, call 47: cycle trace of the convolution kernel with a second stream busy (what the two-stream step does to it)
mkdir -p gpurun_out
for c in "" wgrad conv; do
  echo "#### companion on the other stream: ${c:-none}"
  CMS_TRACE_COMPANION=$c timeout 300 python tools/conv_trace.py "c2 l3" 2>&1 | grep -E "^==|lifetime|prologue|per K step \(mean|CUs used"
done > gpurun_out/r2ap_conv_trace_two_streams.log 2>&1
cat gpurun_out/r2ap_conv_trace_two_streams.log
