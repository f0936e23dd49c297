#!/bin/bash
# round 2, call 17: whole-step A/B of the convolution loader variants (same box, same run)
mkdir -p gpurun_out
for v in 0 43 50 51 0 43; do
  CMS_CONV_DEFAULT_VARIANT=$v timeout 300 python bench.py --workload pascal --no_cpu_baseline --steps 40 --warmup 5 > gpurun_out/r2q_bench_v$v.log 2> gpurun_out/r2q_bench_v$v.err
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/r2q_bench_v%s.log' % v) if l.startswith('{"metric"')][-1])
print('variant', v, 'img/s %.1f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'in-step frac %.4f' % d['roofline']['frac'], 'isolated %.4f' % d['roofline'].get('isolated', {}).get('frac', 0), 'avg launch ms %.4f' % d['roofline']['avg_launch_ms'])
PY
done 2>&1 | tee gpurun_out/r2q_ab.log
