# final validation of the round: full GPU suite, smoke, default bench line (traffic + cpu baseline), kernel stats, CLI smoke
bash tools/gpu_job.sh r04z "pytest:all" smoke
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
( time timeout 1500 python bench.py ) > $OUT/r04z_bench_default.log 2> $OUT/r04z_bench_default.err
grep '^{"metric"' $OUT/r04z_bench_default.log > $OUT/r04z_bench_default.json
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04z_bench_default.json').read())
r=d['roofline']
print('DEFAULT value',d['value'],'ms',d['ms_per_step'],'512x1024',d.get('value_512x1024'))
print('roofline', {k:r.get(k) for k in ('kernel','achieved','peak','frac','traffic','algorithmic_bytes_per_launch','avg_launch_ms')})
print('mixed', r.get('mixed'), 'isolated', r.get('isolated'))
print('cpu_baseline', d.get('cpu_baseline'))
print('also', [(a['name'],a['value']) for a in d.get('also',[])])
PY
bash tools/gpu_job.sh r04z2 "rocprof:--workload pascal --steps 20 --warmup 5 --no_cpu_baseline --traffic omit --no_also"
( timeout 900 bash tools/cli_smoke.sh ) > $OUT/r04z_cli_smoke.log 2>&1; echo "cli_smoke rc=$?"; tail -n 5 $OUT/r04z_cli_smoke.log | cut -c1-200
