#!/bin/bash
# round 2, call 45: in-step A/B of tile shapes and the (now really pinned) fragment double buffer
mkdir -p gpurun_out
run() { label=$1; shift
  env "$@" timeout 300 python bench.py --workload pascal --no_cpu_baseline --steps 40 --warmup 5 $FLAGS > gpurun_out/r2ao_$label.log 2> gpurun_out/r2ao_$label.err
  python - $label <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads([l for l in open('gpurun_out/r2ao_%s.log' % v) if l.startswith('{"metric"')][-1])
    print('%-28s img/s %.1f  ms %.2f  isolated frac %.4f' % (v, d['value'], d['ms_per_step'], d['roofline'].get('isolated', {}).get('frac', 0)))
except Exception as e:
    print(v, 'FAILED', e)
PY
}
FLAGS="" run default A=1
FLAGS="" run variant50_frag_prefetch CMS_CONV_DEFAULT_VARIANT=50
FLAGS="--conv_tile 2256" run tile2256 A=1
FLAGS="--conv_tile 256" run tile256 A=1
FLAGS="" run default_again A=1
