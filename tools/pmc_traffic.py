"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE): tools/gpu_job.sh pmc:<bench args>.
usage: pmc_traffic.py <dir with FETCH_SIZE/ and WRITE_SIZE/ outputs> <out.json>"""
import collections
import csv
import glob
import json
import sys

root, out_path = sys.argv[1], sys.argv[2]
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(list)
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (root, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == c:
                agg[r['Kernel_Name'][:90]].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out.setdefault(k, {})[c] = {'avg_per_launch': sum(v) / len(v), 'launches': len(v), 'sum': sum(v)}
top = sorted(out.items(), key=lambda kv: -kv[1].get('FETCH_SIZE', {}).get('sum', 0))[:30]
json.dump(dict(top), open(out_path, 'w'), indent=1)
for k, v in top[:14]:
    print(k[:70], {c: (round(d['avg_per_launch'], 1), d['launches']) for c, d in v.items()})
