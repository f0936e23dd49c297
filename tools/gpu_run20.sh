#!/bin/bash
export TMPDIR=/tmp
for t in 2256; do
( timeout 600 python bench.py --no_cpu_baseline --no_roofline_events --conv_tile $t ) > gpurun_out/bench_tmp.log 2>&1; echo "tile $t rc=$?"
grep '^{"metric"' gpurun_out/bench_tmp.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')
"
done
