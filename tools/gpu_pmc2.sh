#!/bin/bash
# PMC passes of one convolution shape (conv_igemm_kernel), counters per the guide's slot table; one pass per group.
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
run() { # name counters shape
  rocprofv3 --kernel-trace --pmc $2 -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$1 -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/conv_one.py $3 > $GRAFT_REPO_ROOT/gpurun_out/pmc/$1.log 2>&1
}
for SH in "41 41 512 512 3 4" "41 41 256 1024 1 1" "41 41 256 256 3 2"; do
  tag=$(echo $SH | tr ' ' '_')
  run a_$tag "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "$SH"
  run b_$tag "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "$SH"
  run c_$tag "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "$SH"
  run d_$tag "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "$SH"
done
cd $GRAFT_REPO_ROOT/gpurun_out/pmc
python - <<'PY'
import csv, glob, collections, os, json
out = {}
for d in sorted(glob.glob('[abcd]_*')):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'conv_igemm' in r.get('Kernel_Name', ''):
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        shape = d[2:]
        for k, v in agg.items():
            out.setdefault(shape, {})[k] = sum(v) / len(v)
json.dump(out, open('../pmc_conv.json', 'w'), indent=1)
for s, c in out.items():
    print(s)
    for k in sorted(c):
        print('   %-28s %.4g' % (k, c[k]))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc
