#!/bin/bash
# round 2, call 40: is the two-stream schedule still worth it? single stream with the occupancy-heavy kernel variants
mkdir -p gpurun_out
run() { # label, env..., -- flags
  label=$1; shift
  env "$@" timeout 300 python bench.py --workload pascal --no_cpu_baseline --steps 40 --warmup 5 $FLAGS > gpurun_out/r2aj_$label.log 2> gpurun_out/r2aj_$label.err
  python - $label <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/r2aj_%s.log' % v) if l.startswith('{"metric"')][-1])
print('%-34s img/s %.1f  ms %.2f' % (v, d['value'], d['ms_per_step']))
PY
}
FLAGS="" run overlap_default A=1
FLAGS="--no_overlap" run single_stream A=1
FLAGS="--no_overlap" run single_wgrad2stage CMS_WGRAD_STAGES=2
FLAGS="--no_overlap" run single_wgrad2stage_slab CMS_WGRAD_STAGES=2 CMS_WGRAD_SLAB=1
FLAGS="--no_overlap" run single_slab CMS_WGRAD_SLAB=1
FLAGS="" run overlap_default_again A=1
