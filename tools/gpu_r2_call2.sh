#!/bin/bash
# round 2, GPU call 2: launch programs (tests + bench), pipelined conv variants sweep
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_programs.py -q -m gpu -x > gpurun_out/r2b_programs.log 2>&1; echo "rc=$?" >> gpurun_out/r2b_programs.log
timeout 300 python tools/conv_variants.py > gpurun_out/r2b_conv_variants.log 2>&1; echo "rc=$?" >> gpurun_out/r2b_conv_variants.log
timeout 600 python -m pytest tests -q -m gpu --deselect tests/test_gpu_programs.py > gpurun_out/r2b_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r2b_pytest_gpu.log
timeout 400 python bench.py --no_cpu_baseline > gpurun_out/r2b_bench.log 2> gpurun_out/r2b_bench.err; echo "rc=$?" >> gpurun_out/r2b_bench.err
tail -n 6 gpurun_out/r2b_programs.log; tail -n 20 gpurun_out/r2b_conv_variants.log; tail -n 6 gpurun_out/r2b_pytest_gpu.log; tail -c 600 gpurun_out/r2b_bench.log; tail -n 5 gpurun_out/r2b_bench.err
