#!/bin/bash
# round 2, call 9: K-rotation / staggered-start variants of the default conv kernel; bench line with the empty-queue
# host measurement; the two Pi-model parity tests after the tolerance fix
mkdir -p gpurun_out
export CMS_VARIANTS=0:0,0:20,0:21,0:22,0:23,0:24,0:25
timeout 300 python tools/conv_variants.py > gpurun_out/r2i_conv_variants.log 2>&1; echo "variants rc=$?"
cat gpurun_out/r2i_conv_variants.log
unset CMS_VARIANTS
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pi" > gpurun_out/r2i_pytest_pi.log 2>&1; echo "pi rc=$?"
tail -3 gpurun_out/r2i_pytest_pi.log
timeout 400 python bench.py --no_cpu_baseline --steps 20 --warmup 5 > gpurun_out/r2i_bench.log 2> gpurun_out/r2i_bench.err; echo "bench rc=$?"
cat gpurun_out/r2i_bench.log | cut -c1-1500
