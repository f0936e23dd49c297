#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_executor.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "executor or step or iteration or trainer" ) > gpurun_out/pytest_exec.log 2>&1; echo "pytest rc=$?"
grep -E "^E  |passed|failed|Error" gpurun_out/pytest_exec.log | cut -c1-250 | head -20
for flags in "--no_roofline_events" ""; do
( timeout 600 python bench.py --no_cpu_baseline $flags ) > gpurun_out/bench_tmp.log 2>&1; echo "bench rc=$?"
grep '^{"metric"' gpurun_out/bench_tmp.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', 'enqueue', round(d['config']['host_enqueue_ms_per_step'],2), 'roof', d['roofline'].get('achieved'), d['roofline'].get('isolated',{}).get('achieved'), d['roofline'].get('step_mfma'))
"
done
tail -3 gpurun_out/bench_tmp.log | grep -v '^{"metric"' | cut -c1-300
