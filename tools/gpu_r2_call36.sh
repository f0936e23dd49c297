#!/bin/bash
# round 2, call 36: split-K weight gradient through slabs + reduce instead of atomics: tests, layer list, step A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_hip_engine_parity.py tests/test_gpu_deeplab3plus.py tests/test_gpu_programs.py -m gpu -x -q > gpurun_out/r2ah_pytest.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2ah_pytest.log
for sl in 1 0; do echo "== CMS_WGRAD_SLAB=$sl"; CMS_WGRAD_SLAB=$sl timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r2ah_wgrad_bench.log; cat gpurun_out/r2ah_wgrad_bench.log
for sl in 1 0 1 0; do
CMS_WGRAD_SLAB=$sl timeout 300 python bench.py --workload pascal --no_cpu_baseline --steps 40 --warmup 5 > gpurun_out/r2ah_bench_s$sl.log 2> gpurun_out/r2ah_bench_s$sl.err
python - $sl <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/r2ah_bench_s%s.log' % v) if l.startswith('{"metric"')][-1])
print('wgrad slab', v, 'img/s %.1f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'loss', d['config']['last_losses']['sup_loss'])
PY
done
