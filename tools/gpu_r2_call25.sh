#!/bin/bash
# round 2, call 25: balanced launch (two tile shapes in one kernel): tests, isolated sweep A/B, step A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_hip_engine_parity.py tests/test_gpu_programs.py -m gpu -x -q > gpurun_out/r2x_pytest.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2x_pytest.log
export CMS_VARIANTS=0:0,128:0
for mx in 1 0; do echo "== CMS_CONV_MIXED=$mx"; CMS_CONV_MIXED=$mx timeout 300 python tools/conv_variants.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r2x_conv_variants.log; cat gpurun_out/r2x_conv_variants.log
unset CMS_VARIANTS
for mx in 1 0 1 0; do
CMS_CONV_MIXED=$mx timeout 300 python bench.py --workload pascal --no_cpu_baseline --steps 40 --warmup 5 > gpurun_out/r2x_bench_m$mx.log 2> gpurun_out/r2x_bench_m$mx.err
python - $mx <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/r2x_bench_m%s.log' % v) if l.startswith('{"metric"')][-1])
print('mixed', v, 'img/s %.1f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'in-step frac %.4f' % d['roofline']['frac'], 'isolated %.4f' % d['roofline'].get('isolated', {}).get('frac', 0))
PY
done
