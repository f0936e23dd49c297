#!/bin/bash
# PMC passes of the weight-gradient kernel (two layer shapes) + the single-stream reference point of the bench.
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
run() { rocprofv3 --kernel-trace --pmc $2 -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$1 -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/wgrad_one.py $3 > $GRAFT_REPO_ROOT/gpurun_out/pmc/$1.log 2>&1; }
for SH in "41 41 256 256 3 2" "41 41 1024 256 1 1"; do
  tag=$(echo $SH | tr ' ' '_')
  run a_$tag "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "$SH"
  run b_$tag "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "$SH"
  run c_$tag "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "$SH"
done
cd $GRAFT_REPO_ROOT/gpurun_out/pmc
python - <<'PY'
import csv, glob, collections, os, json
out = {}
for d in sorted(glob.glob('[abc]_*')):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'conv_wgrad' in r.get('Kernel_Name', ''):
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            out.setdefault(d[2:], {})[k] = sum(v) / len(v)
json.dump(out, open('../pmc_wgrad.json', 'w'), indent=1)
for s, c in out.items():
    wc = c.get('SQ_WAVE_CYCLES', 1)
    print(s, 'mfma_busy_frac %.3f' % (c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024.0 / (c.get('GRBM_GUI_ACTIVE', 1) / 8.0)),
          'wait_any %.2f wait_inst %.2f active %.2f' % (c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_ANY', 0) / wc),
          'valu/mfma %.1f lds/mfma %.2f' % (c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_INSTS_MFMA', 1), 1), c.get('SQ_INSTS_LDS', 0) / max(c.get('SQ_INSTS_MFMA', 1), 1)),
          'l2hit %.2f' % (c.get('TCC_HIT_sum', 0) / max(c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0), 1)), 'bank_conflict %.3g' % c.get('SQ_LDS_BANK_CONFLICT', 0))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc
cd $GRAFT_REPO_ROOT
( timeout 600 python bench.py --no_cpu_baseline --no_overlap --no_roofline_events ) > gpurun_out/bench_tmp.log 2>&1
grep '^{"metric"' gpurun_out/bench_tmp.log | cut -c1-200
