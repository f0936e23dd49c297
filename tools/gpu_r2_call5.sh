#!/bin/bash
# round 2, GPU call 5: U-Nets, N1, VAT tests; VAT throughput (DeepLab on the fp32 HIP engine, DenseNet-161 U-Net)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_unets.py tests/test_gpu_miou_training.py tests/test_gpu_vat.py tests/test_gpu_hip_engine_parity.py tests/test_gpu_executor.py -q -m gpu -s > gpurun_out/r2e_new.log 2>&1; echo "rc=$?" >> gpurun_out/r2e_new.log
timeout 300 python tools/vat_bench.py deeplab > gpurun_out/r2e_vat_deeplab.log 2>&1
timeout 300 python tools/vat_bench.py denseunet > gpurun_out/r2e_vat_denseunet.log 2>&1
grep -E "passed|failed|rc=|^FAILED|N1 |configs\[|unet" gpurun_out/r2e_new.log | cut -c1-300 | tail -n 20; tail -n 2 gpurun_out/r2e_vat_deeplab.log gpurun_out/r2e_vat_denseunet.log
