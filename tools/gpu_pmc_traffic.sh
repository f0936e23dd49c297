#!/bin/bash
# HBM traffic of the bench kernels from the TCC memory-side counters, one counter per pass (MI355X_MICROARCH.md,
# "HBM" + "rocprofv3 PMC slots"): FETCH_SIZE and WRITE_SIZE cannot share a pass. Serial streams so that a
# dispatch's counters are its own. Output: gpurun_out/pmc_traffic.json (per-kernel averages per launch).
mkdir -p gpurun_out/pmc_traffic
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic/$c -o p --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_overlap --no_roofline_events \
    > $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic/$c.log 2>&1
  echo "$c rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    fs = glob.glob('gpurun_out/pmc_traffic/%s/**/*counter_collection.csv' % c, recursive=True)
    agg = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == c:
                agg[r['Kernel_Name'][:90]].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out.setdefault(k, {})[c] = {'avg_per_launch': sum(v) / len(v), 'launches': len(v), 'sum': sum(v)}
top = sorted(out.items(), key=lambda kv: -kv[1].get('FETCH_SIZE', {}).get('sum', 0))[:25]
json.dump(dict(top), open('gpurun_out/pmc_traffic.json', 'w'), indent=1)
for k, v in top[:12]:
    print(k[:70], {c: (round(d['avg_per_launch'], 1), d['launches']) for c, d in v.items()})
PY
rm -rf gpurun_out/pmc_traffic/FETCH_SIZE gpurun_out/pmc_traffic/WRITE_SIZE
