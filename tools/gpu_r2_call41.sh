#!/bin/bash
# round 2, call 41: weight gradients on 1 / 2 / 3 streams with the direct-to-LDS kernels
mkdir -p gpurun_out
for ws in 1 2 3 1; do
  timeout 300 python bench.py --workload pascal --no_cpu_baseline --steps 40 --warmup 5 --wgrad_streams $ws > gpurun_out/r2ak_$ws.log 2> gpurun_out/r2ak_$ws.err
  python - $ws <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/r2ak_%s.log' % v) if l.startswith('{"metric"')][-1])
print('wgrad streams %s  img/s %.1f  ms %.2f' % (v, d['value'], d['ms_per_step']))
PY
done
