#!/bin/bash
# round 2, call 21: direct-to-LDS loads issued from inline asm (invisible to hipcc's vmcnt(0)-before-ds_read): conv
# rings and the two-stage wgrad become real pipelines. Tests, sweeps, traces, step bench.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_hip_engine_parity.py tests/test_gpu_deeplab3plus.py tests/test_gpu_programs.py -m gpu -x -q > gpurun_out/r2t_pytest.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2t_pytest.log
export CMS_VARIANTS=0:0,0:43,0:50,0:51,0:52,0:53,0:54
timeout 300 python tools/conv_variants.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2t_conv_variants.log; cat gpurun_out/r2t_conv_variants.log
unset CMS_VARIANTS
timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2t_wgrad_bench.log; cat gpurun_out/r2t_wgrad_bench.log
timeout 200 python tools/wgrad_trace.py l3 2>&1 | grep -v amdgpu.ids > gpurun_out/r2t_wgrad_trace.log; grep -E "^==|prologue|per stage" gpurun_out/r2t_wgrad_trace.log
for v in 0 50 51; do
CMS_CONV_DEFAULT_VARIANT=$v timeout 300 python bench.py --workload pascal --no_cpu_baseline --steps 40 --warmup 5 > gpurun_out/r2t_bench_v$v.log 2> gpurun_out/r2t_bench_v$v.err
python - $v <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/r2t_bench_v%s.log' % v) if l.startswith('{"metric"')][-1])
print('conv variant', v, 'img/s %.1f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'in-step frac %.4f' % d['roofline']['frac'], 'isolated %.4f' % d['roofline'].get('isolated', {}).get('frac', 0))
PY
done
