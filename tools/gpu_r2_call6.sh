#!/bin/bash
# round 2, GPU call 6: weight gradients on 1 / 2 / 3 side streams (bench A/B), dist + program + executor tests
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_programs.py tests/test_gpu_dist.py tests/test_gpu_executor.py tests/test_gpu_unets.py -q -m gpu > gpurun_out/r2f_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2f_tests.log
for k in 1 2 3 2 1 3; do
  timeout 200 python bench.py --workload pascal --no_cpu_baseline --wgrad_streams $k --steps 40 > gpurun_out/r2f_bench_ws$k.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/r2f_bench_ws$k.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('wgrad_streams=$k', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')
else:
    print('wgrad_streams=$k: no result'); print(open('gpurun_out/r2f_bench_ws$k.log').read()[-1500:])
PY
done | tee gpurun_out/r2f_ws_summary.txt
for k in 2 3; do
  timeout 200 python bench.py --workload cityscapes --no_cpu_baseline --wgrad_streams $k --steps 30 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('cityscapes wgrad_streams=$k', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')
" | tee -a gpurun_out/r2f_ws_summary.txt
done
grep -E "passed|failed|rc=|^FAILED" gpurun_out/r2f_tests.log | tail -n 5
