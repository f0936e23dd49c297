#!/bin/bash
# round 2, call 16: buffer addressing as the default loader; ring variants with it; tests; bench
mkdir -p gpurun_out
export CMS_VARIANTS=0:0,0:43,0:50,0:51,0:52,0:53,0:54
timeout 300 python tools/conv_variants.py > gpurun_out/r2p_conv_variants.log 2>&1; echo "variants rc=$?"
cat gpurun_out/r2p_conv_variants.log
unset CMS_VARIANTS
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_programs.py tests/test_gpu_hip_engine_parity.py tests/test_gpu_deeplab3plus.py -m gpu -x -q > gpurun_out/r2p_pytest.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2p_pytest.log
timeout 400 python bench.py --no_cpu_baseline --steps 30 --warmup 5 > gpurun_out/r2p_bench.log 2> gpurun_out/r2p_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r2p_bench.log') if l.startswith('{"metric"')][-1])
print(d['value'], d['ms_per_step'], d.get('value_512x1024'), d['roofline']['frac'], d['roofline'].get('isolated', {}).get('frac'))
PY
