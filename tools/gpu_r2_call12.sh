#!/bin/bash
# round 2, call 12: async scale / bias staging + wgrad scale in LDS: conv / wgrad / engine tests, trace, bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_programs.py tests/test_gpu_hip_engine_parity.py tests/test_gpu_deeplab3plus.py -m gpu -x -q > gpurun_out/r2l_pytest.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2l_pytest.log
timeout 300 python tools/conv_trace.py > gpurun_out/r2l_conv_trace.log 2>&1; echo "trace rc=$?"
grep -E "^==|prologue|per K step \(mean" gpurun_out/r2l_conv_trace.log
timeout 300 python tools/wgrad_bench.py > gpurun_out/r2l_wgrad_bench.log 2>&1; cat gpurun_out/r2l_wgrad_bench.log
timeout 400 python bench.py --no_cpu_baseline --steps 30 --warmup 5 > gpurun_out/r2l_bench.log 2> gpurun_out/r2l_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r2l_bench.log') if l.startswith('{"metric"')][-1])
print(d['value'], d['ms_per_step'], d.get('value_512x1024'), d['roofline']['frac'], d['roofline'].get('isolated', {}).get('frac'))
PY
