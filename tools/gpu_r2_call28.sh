#!/bin/bash
# round 2, call 28: stem weight gradient on the matrix cores: tests, step A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_stem.py tests/test_gpu_hip_engine_parity.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2aa_pytest.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r2aa_pytest.log
for mf in 1 0 1 0; do
CMS_STEM_MFMA=$mf timeout 300 python bench.py --workload pascal --no_cpu_baseline --steps 40 --warmup 5 > gpurun_out/r2aa_bench_m$mf.log 2> gpurun_out/r2aa_bench_m$mf.err
python - $mf <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/r2aa_bench_m%s.log' % v) if l.startswith('{"metric"')][-1])
print('stem mfma', v, 'img/s %.1f' % d['value'], 'ms %.2f' % d['ms_per_step'], 'loss', d['config']['last_losses'])
PY
done
