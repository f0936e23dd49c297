#!/usr/bin/env python
"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE): tools/gpu_job.sh mfma:<bench args>.
usage: pmc_mfma.py <dir with the counter_collection csv> <out.json>
util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): the fraction of the chip's MFMA issue cycles (256 CUs x 4 SIMDs) the kernel
used while it ran. SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md; checked:
39.56 M per conv8 launch = 1.236 M MFMAs = the launch's 40.3 GFLOP / 32 768); GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (1.23 M per
launch against ~0.14 M shader cycles of its ~60 us). Run the bench with --no_overlap: serial streams, so a dispatch's counters are its own."""
import collections
import csv
import glob
import json
import sys

root, out_path = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('%s/**/*counter_collection.csv' % root, recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in agg.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in cs or 'GRBM_GUI_ACTIVE' not in cs:
        continue
    busy, act = sum(cs['SQ_VALU_MFMA_BUSY_CYCLES']), sum(cs['GRBM_GUI_ACTIVE'])
    if busy <= 0 or act <= 0:
        continue
    out[k] = {'launches': len(cs['GRBM_GUI_ACTIVE']), 'mfma_busy_cycles_per_launch': busy / len(cs['SQ_VALU_MFMA_BUSY_CYCLES']),
              'gui_active_cycles_per_launch': act / len(cs['GRBM_GUI_ACTIVE']), 'mfma_util_of_chip': busy / (act / 8.0 * 1024.0)}
top = sorted(out.items(), key=lambda kv: -kv[1]['mfma_busy_cycles_per_launch'] * kv[1]['launches'])[:16]
json.dump(dict(top), open(out_path, 'w'), indent=1)
for k, v in top:
    print(k[:70].ljust(70), 'launches', v['launches'], 'MFMA util of the chip while running %.3f' % v['mfma_util_of_chip'])
