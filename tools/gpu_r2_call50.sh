#!/bin/bash
# round 2, call 50: per-layer tile rules inside the step (fewer, bigger workgroups for the short-K expansions?)
mkdir -p gpurun_out
run() { label=$1; shift
  timeout 300 python bench.py --workload pascal --no_cpu_baseline --steps 40 --warmup 5 "$@" > gpurun_out/r2ar_$label.log 2> gpurun_out/r2ar_$label.err
  python - $label <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads([l for l in open('gpurun_out/r2ar_%s.log' % v) if l.startswith('{"metric"')][-1])
    print('%-28s img/s %.1f  ms %.2f' % (v, d['value'], d['ms_per_step']))
except Exception as e:
    print(v, 'FAILED', e)
PY
}
run default
run rule_1024_2256 --tile_rule 1024:2256
run rule_1024_256 --tile_rule 1024:256
run rule_1024_2048_2256 --tile_rule 1024:2256,2048:2256
run rule_1024_64 --tile_rule 1024:64
run default_again
