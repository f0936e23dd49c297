#!/bin/bash
# round 2, call 30: recorded passes for the DeepLab v3+ executor: tests, cfg 4 bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_deeplab3plus.py -m gpu -x -q > gpurun_out/r2ac_pytest.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r2ac_pytest.log
timeout 300 python bench.py --workload pascal_v3plus --no_cpu_baseline --steps 20 > gpurun_out/r2ac_v3.log 2> gpurun_out/r2ac_v3.err; echo "bench rc=$?"; tail -3 gpurun_out/r2ac_v3.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2ac_v3.log") if l.startswith('{"metric"')][-1])
print(d["value"], d["ms_per_step"], d["config"].get("host_enqueue_ms_per_step"), d["config"].get("host_enqueue_ms_per_step_empty_queue"), d['config'].get('last_losses'))
PY
