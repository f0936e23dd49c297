#!/bin/bash
# round 2, call 22: wgrad with fragment double buffering, one vs two direct-to-LDS stages: tests, sweep, step A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_hip_engine_parity.py tests/test_gpu_deeplab3plus.py -m gpu -x -q > gpurun_out/r2u_pytest.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2u_pytest.log
for st in 1 2; do
  echo "== CMS_WGRAD_STAGES=$st"
  CMS_WGRAD_STAGES=$st timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r2u_wgrad_bench.log 2>&1
cat gpurun_out/r2u_wgrad_bench.log
CMS_WGRAD_STAGES=2 timeout 200 python tools/wgrad_trace.py l3 2>&1 | grep -v amdgpu.ids > gpurun_out/r2u_wgrad_trace.log; grep -E "^==|prologue|per stage" gpurun_out/r2u_wgrad_trace.log
for st in 1 2 1; do
CMS_WGRAD_STAGES=$st timeout 300 python bench.py --workload pascal --no_cpu_baseline --steps 40 --warmup 5 > gpurun_out/r2u_bench_s$st.log 2> gpurun_out/r2u_bench_s$st.err
python - $st <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/r2u_bench_s%s.log' % v) if l.startswith('{"metric"')][-1])
print('wgrad stages', v, 'img/s %.1f' % d['value'], 'ms %.2f' % d['ms_per_step'])
PY
done
