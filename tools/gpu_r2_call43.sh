#!/bin/bash
# round 2, call 43: slice-count model for the weight gradient: layer list, cfg 2 and cfg 4 step A/B
mkdir -p gpurun_out
for md in 1 0; do echo "== CMS_WGRAD_MODEL=$md"; CMS_WGRAD_MODEL=$md timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-32,70-; done > gpurun_out/r2am_wgrad_bench.log; cat gpurun_out/r2am_wgrad_bench.log
for md in 1 0 1 0; do
for wl in pascal pascal_v3plus; do
CMS_WGRAD_MODEL=$md timeout 300 python bench.py --workload $wl --no_cpu_baseline --steps 30 --warmup 5 > gpurun_out/r2am_bench_${wl}_$md.log 2> gpurun_out/r2am_bench_${wl}_$md.err
python - $md $wl <<'PY'
import json, sys
v, wl = sys.argv[1:3]
d = json.loads([l for l in open('gpurun_out/r2am_bench_%s_%s.log' % (wl, v)) if l.startswith('{"metric"')][-1])
print('model', v, '%-14s' % wl, 'img/s %.1f' % d['value'], 'ms %.2f' % d['ms_per_step'])
PY
done
done
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_hip_engine_parity.py -m gpu -x -q 2>&1 | tail -2
