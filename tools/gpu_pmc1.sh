#!/bin/bash
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
export PMC_KERNEL=${PMC_KERNEL:-conv_igemm}
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "^\s*(Name|Counter_Name)\s*:\s*[A-Za-z0-9_]+" | awk '{print $NF}' | sort -u > $GRAFT_REPO_ROOT/gpurun_out/pmc/counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/pmc/counters.txt
run() { # name, counters..., -- args
  name=$1; shift
  rocprofv3 --kernel-trace --pmc $1 -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$name -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/${PMC_SCRIPT:-conv_one.py} $2 > $GRAFT_REPO_ROOT/gpurun_out/pmc/$name.log 2>&1
}
SHAPE="${PMC_SHAPE:-41 41 512 512 3 4}"
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "$SHAPE"
run b "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "$SHAPE"
run c "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "$SHAPE"
cd $GRAFT_REPO_ROOT/gpurun_out/pmc
for d in a b c; do f=$(find $d -name "*counter_collection.csv" | head -1); echo "== $d $f"; python - "$f" <<'PY'
import csv, sys, collections, os
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if os.environ.get('PMC_KERNEL', 'conv_igemm') in r.get('Kernel_Name', ''):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k, sum(v) / len(v), len(v))
PY
done
tail -3 a.log
