#!/bin/bash
export TMPDIR=/tmp
cd /tmp && ( timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload pascal_v3plus --steps 4 --warmup 4 --no_cpu_baseline ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof_tmp/bench_results.db 45 > gpurun_out/kernel_stats_v3.csv 2>&1
rm -rf gpurun_out/prof_tmp
python - <<PY
import csv
rows=list(csv.reader(open("gpurun_out/kernel_stats_v3.csv")))
for r in rows[1:32]:
    print(r[0][:70].ljust(70), r[1].rjust(6), r[2].rjust(12), r[3].rjust(10), r[-1].rjust(6))
PY
