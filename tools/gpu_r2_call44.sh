#!/bin/bash
# round 2, call 44: <W, G> side output of the weight gradient with batched loads + DPP row sums: tests, cfg 4 bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_deeplab3plus.py -m gpu -x -q > gpurun_out/r2an_pytest.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2an_pytest.log
for i in 1 2; do
timeout 300 python bench.py --workload pascal_v3plus --no_cpu_baseline --steps 30 --warmup 5 > gpurun_out/r2an_v3.log 2> gpurun_out/r2an_v3.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r2an_v3.log') if l.startswith('{"metric"')][-1])
print('v3+ img/s %.1f ms %.2f' % (d['value'], d['ms_per_step']), d['config']['last_losses'])
PY
done
