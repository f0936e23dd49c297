#!/bin/bash
# round 2 validation (second pass, after the trace-driven kernel work) on a GPU box: smoke, the whole -m gpu suite, the default bench line (both shapes + CPU baselines),
# rocprofv3 kernel trace of the bench, TCC traffic counters (separate FETCH_SIZE / WRITE_SIZE passes)
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python __graft_entry__.py smoke ) > gpurun_out/r2z_smoke.log 2>&1; echo "smoke rc=$?" | tee gpurun_out/r2z_summary.log
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r2z_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2z_summary.log
( time timeout 600 python bench.py ) > gpurun_out/r2z_bench.log 2> gpurun_out/r2z_bench.err; echo "bench rc=$?" | tee -a gpurun_out/r2z_summary.log
cd /tmp && ( timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2z_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no_cpu_baseline ) > $GRAFT_REPO_ROOT/gpurun_out/r2z_rocprof.log 2>&1; echo "rocprof rc=$?" | tee -a $GRAFT_REPO_ROOT/gpurun_out/r2z_summary.log
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/r2z_prof/bench_results.db 60 > gpurun_out/r2z_kernel_stats.csv 2>> gpurun_out/r2z_rocprof.log
rm -rf gpurun_out/r2z_prof
mkdir -p gpurun_out/r2z_pmc
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/r2z_pmc/$c -o p --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --workload pascal --steps 2 --warmup 1 --no_cpu_baseline --no_overlap --no_roofline_events \
    > $GRAFT_REPO_ROOT/gpurun_out/r2z_pmc/$c.log 2>&1
  echo "$c rc=$?" | tee -a $GRAFT_REPO_ROOT/gpurun_out/r2z_summary.log
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    fs = glob.glob('gpurun_out/r2z_pmc/%s/**/*counter_collection.csv' % c, recursive=True)
    agg = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == c:
                agg[r['Kernel_Name'][:90]].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out.setdefault(k, {})[c] = {'avg_per_launch': sum(v) / len(v), 'launches': len(v), 'sum': sum(v)}
top = sorted(out.items(), key=lambda kv: -kv[1].get('FETCH_SIZE', {}).get('sum', 0))[:30]
json.dump(dict(top), open('gpurun_out/r2z_pmc_traffic_per_kernel.json', 'w'), indent=1)
for k, v in top[:14]:
    print(k[:70], {c: (round(d['avg_per_launch'], 1), d['launches']) for c, d in v.items()})
PY
rm -rf gpurun_out/r2z_pmc/FETCH_SIZE gpurun_out/r2z_pmc/WRITE_SIZE
tail -n 3 gpurun_out/r2z_smoke.log
grep -E "^FAILED|passed|failed" gpurun_out/r2z_pytest_gpu.log | tail -n 12
grep '^{"metric"' gpurun_out/r2z_bench.log | cut -c1-300
head -n 8 gpurun_out/r2z_kernel_stats.csv | cut -c1-150
