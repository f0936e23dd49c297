#!/usr/bin/env python
"""Cycle trace of the weight-gradient kernel (csrc/conv.hip conv_wgrad_kernel; stamps switched on by
cms_conv_set_trace). Wave 0 of every workgroup stamps s_memtime per 64-pixel stage:

   t0 top | barrier | t1 | wait global loads + ds_write stage | t2 | barrier | t3 | issue next loads | t4 | 32 tr-reads + 16 MFMA | t5
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cutmix_semisup_seg_amd import ops
from cutmix_semisup_seg_amd._lib import lib

DEV = 'cuda:0'
TD = 512
N = 20
SHAPES = [('l3 1x1 1024->256', 41, 41, 1024, 256, 1, 1), ('l3 3x3d2 256->256', 41, 41, 256, 256, 3, 2),
          ('l3 1x1 256->1024', 41, 41, 256, 1024, 1, 1), ('l4 3x3d4 512->512', 41, 41, 512, 512, 3, 4),
          ('l2 1x1 128->512', 41, 41, 128, 512, 1, 1)]
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if any(a in s[0] for a in sys.argv[1:])]
for name, H, W, Cin, Cout, k, dil in SHAPES:
    g = torch.Generator(device=DEV).manual_seed(0)
    pad = dil * (k - 1) // 2
    x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
    du = torch.randn(N, H, W, Cout, generator=g, device=DEV).bfloat16()
    dw = torch.zeros(k * k, Cout, Cin, device=DEV)
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    taps = ops.conv_taps(k, k, dil, pad)
    for _ in range(3):
        ops.conv_wgrad(du, x, taps, dw, scale=scale)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv_wgrad(du, x, taps, dw, scale=scale)
    e1.record()
    torch.cuda.synchronize()
    t_plain = e0.elapsed_time(e1) * 1e3 / 5
    nwg = 4096
    buf = torch.zeros(nwg * TD, dtype=torch.int32, device=DEV)
    lib.cms_conv_set_trace(buf.data_ptr(), nwg)
    ops.conv_wgrad(du, x, taps, dw, scale=scale)
    torch.cuda.synchronize()
    lib.cms_conv_set_trace(None, 0)
    tr = buf.cpu().numpy().view(np.uint32).reshape(nwg, TD).astype(np.int64)
    used = tr[:, 7] > 0
    tr = tr[used]
    n = len(tr)
    steps = int(tr[:, 8].min())
    hw, xcc = tr[:, 0], tr[:, 1] & 0xf
    cu_key = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)
    st = tr[:, 16:16 + 6 * steps].reshape(n, steps, 6)
    bar1 = st[:, :, 1] - st[:, :, 0]
    store = st[:, :, 2] - st[:, :, 1]
    bar2 = st[:, :, 3] - st[:, :, 2]
    issue = st[:, :, 4] - st[:, :, 3]
    mfma = st[:, :, 5] - st[:, :, 4]
    step = st[:, 1:, 0] - st[:, :-1, 0] if steps > 1 else st[:, :, 5] - st[:, :, 0]
    life = tr[:, 7]
    print('== {}: {} workgroups, {}..{} stages each, {:.1f} us'.format(name, n, steps, int(tr[:, 8].max()), t_plain))
    print('   lifetime cycles: mean {:.0f}  p10 {:.0f}  p90 {:.0f}  max {:.0f}'.format(
        life.mean(), np.percentile(life, 10), np.percentile(life, 90), life.max()))
    print('   prologue {:.0f} | pixel loop {:.0f} | epilogue issue {:.0f} | atomics acknowledged {:.0f}'.format(
        tr[:, 4].mean(), (tr[:, 5] - tr[:, 4]).mean(), (tr[:, 6] - tr[:, 5]).mean(), (tr[:, 7] - tr[:, 6]).mean()))
    print('   per stage (mean cycles): step {:.0f} = barrier {:.0f} + wait loads & ds_write {:.0f} + barrier {:.0f} + issue next {:.0f}'
          ' + tr-reads & MFMA {:.0f}   (MFMA floor 512)'.format(step.mean(), bar1.mean(), store.mean(), bar2.mean(), issue.mean(),
                                                              mfma.mean()))
    keys, counts = np.unique(cu_key, return_counts=True)
    print('   CUs used {}  workgroups per CU: min {} / mean {:.2f} / max {}   histogram {}'.format(
        len(keys), counts.min(), counts.mean(), counts.max(),
        {int(a): int(b) for a, b in zip(*np.unique(counts, return_counts=True))}))
    sys.stdout.flush()
