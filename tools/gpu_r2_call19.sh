#!/bin/bash
# round 2, call 19: wgrad epilogue through LDS (rotated rows) + split heuristic: tests, bench, trace, step bench
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_hip_engine_parity.py tests/test_gpu_deeplab3plus.py tests/test_gpu_programs.py -m gpu -x -q > gpurun_out/r2s_pytest.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2s_pytest.log
timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2s_wgrad_bench.log; cat gpurun_out/r2s_wgrad_bench.log
timeout 200 python tools/wgrad_trace.py l3 l2 2>&1 | grep -v amdgpu.ids > gpurun_out/r2s_wgrad_trace.log; grep -E "^==|prologue|per stage" gpurun_out/r2s_wgrad_trace.log
for dma in 1 0; do
CMS_WGRAD_DMA=$dma timeout 300 python bench.py --workload pascal --no_cpu_baseline --steps 40 --warmup 5 > gpurun_out/r2s_bench_dma$dma.log 2> gpurun_out/r2s_bench_dma$dma.err
python - $dma <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/r2s_bench_dma%s.log' % v) if l.startswith('{"metric"')][-1])
print('wgrad dma', v, 'img/s %.1f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'in-step frac %.4f' % d['roofline']['frac'])
PY
done
