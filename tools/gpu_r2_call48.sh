#!/bin/bash
# round 2, call 48: rocprofv3 kernel stats of the headline workload alone (pascal), to set beside bench.py's avg_launch_ms
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2aq_prof -o p -- python $GRAFT_REPO_ROOT/bench.py --workload pascal --steps 20 --warmup 2 --no_cpu_baseline --timed_only > $GRAFT_REPO_ROOT/gpurun_out/r2aq_bench.log 2> $GRAFT_REPO_ROOT/gpurun_out/r2aq_prof.log; cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/r2aq_prof/p_results.db 40 > gpurun_out/r2aq_kernel_stats_pascal.csv 2>> gpurun_out/r2aq_prof.log; rm -rf gpurun_out/r2aq_prof
cut -c1-140 gpurun_out/r2aq_kernel_stats_pascal.csv | head -14
python - <<'PY'
import json, csv
d = json.loads([l for l in open('gpurun_out/r2aq_bench.log') if l.startswith('{"metric"')][-1])
print('bench (under rocprof): img/s %.1f  conv avg_launch_ms %.4f over %d timed launches' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['launches_timed']))
rows = list(csv.DictReader(open('gpurun_out/r2aq_kernel_stats_pascal.csv')))
tot = sum(float(r['total_us']) for r in rows if 'conv_igemm' in r['kernel']); n = sum(int(r['calls']) for r in rows if 'conv_igemm' in r['kernel'])
print('rocprofv3: conv_igemm kernels (balanced + plain + 64-channel tile): %d launches, average %.2f us' % (n, tot / n))
PY
