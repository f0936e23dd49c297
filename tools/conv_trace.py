#!/usr/bin/env python
"""Cycle trace of the default convolution kernel (cms_conv_igemm variant 30, csrc/conv.hip): where a workgroup's
cycles go. Wave 0 of every workgroup stamps s_memtime at each phase of a K step:

    t0 loop top | wait own loads (vmcnt 0) | t1 | barrier | t2 | 16 ds_read + 16 MFMA | t3 | barrier | t4 | issue next stage | t5

The tool prints, per layer shape: workgroup lifetime, prologue / K loop / epilogue split, the mean cycles of each
phase of a K step, and how the workgroups that shared one CU were placed in time (s_memrealtime, 100 MHz)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cutmix_semisup_seg_amd import ops
from cutmix_semisup_seg_amd._lib import lib

DEV = 'cuda:0'
TD = 512
SHAPES = [
    # name, N, H, W, Cin, Cout, k, dil, residual
    ('c2 l3 1x1 1024->256', 20, 41, 41, 1024, 256, 1, 1, False),
    ('c2 l3 3x3d2 256->256', 20, 41, 41, 256, 256, 3, 2, False),
    ('c2 l3 1x1 256->1024+res', 20, 41, 41, 256, 1024, 1, 1, True),
    ('c2 l4 3x3d4 512->512', 20, 41, 41, 512, 512, 3, 4, False),
    ('c3 l3 3x3d2 256->256', 8, 65, 129, 256, 256, 3, 2, False),
]
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if any(a in s[0] for a in sys.argv[1:])]

for name, N, H, W, Cin, Cout, k, dil, use_res in SHAPES:
    g = torch.Generator(device=DEV).manual_seed(0)
    pad = dil * (k - 1) // 2
    x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
    wp = (torch.randn(k * k, Cout, Cin, generator=g, device=DEV) * (2.0 / (Cin * k * k)) ** 0.5).bfloat16()
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    bias = torch.randn(Cout, generator=g, device=DEV) * 0.1
    res = torch.randn(N, H, W, Cout, generator=g, device=DEV).bfloat16() if use_res else None
    taps = ops.conv_taps(k, k, dil, pad)
    out = torch.empty(N, H, W, Cout, dtype=torch.bfloat16, device=DEV)
    nwg = ((N * H * W + 127) // 128) * (Cout // 128)
    buf = torch.zeros(nwg * TD, dtype=torch.int32, device=DEV)
    for _ in range(3):
        ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, res=res, relu=True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, res=res, relu=True, out=out)
    e1.record()
    torch.cuda.synchronize()
    t_plain = e0.elapsed_time(e1) * 1e3
    companion = os.environ.get('CMS_TRACE_COMPANION', '')       # 'wgrad' / 'conv': a second stream keeps the machine busy
    if companion:
        side = torch.cuda.Stream()
        gw = torch.Generator(device=DEV).manual_seed(1)
        cx = torch.randn(20, 41, 41, 512, generator=gw, device=DEV).bfloat16()
        cdu = torch.randn(20, 41, 41, 512, generator=gw, device=DEV).bfloat16()
        ctaps = ops.conv_taps(3, 3, 4, 4)
        cdw = torch.zeros(9, 512, 512, device=DEV)
        cw = (torch.randn(9, 512, 512, generator=gw, device=DEV) * 0.02).bfloat16()
        cout = torch.empty(20, 41, 41, 512, dtype=torch.bfloat16, device=DEV)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(6):                                  # ~1.2 ms of work: the traced launch runs inside it
                if companion == 'wgrad':
                    ops.conv_wgrad(cdu, cx, ctaps, cdw)
                else:
                    ops.conv_igemm(cx, cw, ctaps, relu=True, out=cout)
    lib.cms_conv_set_trace(buf.data_ptr(), nwg)
    e0.record()
    ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, res=res, relu=True, out=out,
                   variant=int(os.environ.get('CMS_TRACE_VARIANT', '30')))
    e1.record()
    torch.cuda.synchronize()
    t_traced = e0.elapsed_time(e1) * 1e3
    lib.cms_conv_set_trace(None, 0)
    tr = buf.cpu().numpy().view(np.uint32).reshape(nwg, TD).astype(np.int64)
    steps = int(tr[0, 8])
    hw, xcc = tr[:, 0], tr[:, 1] & 0xf
    cu = (hw >> 8) & 0xf
    se = (hw >> 13) & 0x7
    sh = (hw >> 12) & 0x1
    cu_key = xcc * 1000 + se * 100 + sh * 16 + cu
    rt = tr[:, 2] + (tr[:, 3] << 32)                    # 100 MHz ticks
    life = tr[:, 7]
    st = tr[:, 16:16 + 6 * steps].reshape(nwg, steps, 6)
    wait_own = st[:, :, 1] - st[:, :, 0]
    bar1 = st[:, :, 2] - st[:, :, 1]
    mfma = st[:, :, 3] - st[:, :, 2]
    bar2 = st[:, :, 4] - st[:, :, 3]
    issue = st[:, :, 5] - st[:, :, 4]
    step = st[:, 1:, 0] - st[:, :-1, 0] if steps > 1 else st[:, :, 5] - st[:, :, 0]
    # clock estimate: cycles of the longest-lived workgroup over its realtime span is not available per WG end, so use
    # the kernel: cycles from first start to last end / event time
    print('== {}: {} workgroups, {} K steps, event time {:.1f} us plain / {:.1f} us traced'.format(
        name, nwg, steps, t_plain, t_traced))
    print('   lifetime cycles: mean {:.0f}  p10 {:.0f}  p90 {:.0f}  max {:.0f}'.format(
        life.mean(), np.percentile(life, 10), np.percentile(life, 90), life.max()))
    print('   prologue {:.0f} | K loop {:.0f} | epilogue math {:.0f} | stores {:.0f}   (mean cycles)'.format(
        tr[:, 4].mean(), (tr[:, 5] - tr[:, 4]).mean(), (tr[:, 6] - tr[:, 5]).mean(), (tr[:, 7] - tr[:, 6]).mean()))
    print('   prologue split: tables written at {:.0f} | barrier + loader geometry until {:.0f} | first tap offsets until {:.0f} | '
          'first stage issued at {:.0f}'.format(tr[:, 13].mean(), tr[:, 14].mean(), tr[:, 15].mean(), tr[:, 4].mean()))
    print('   per K step (mean cycles): step {:.0f} = wait-own-loads {:.0f} + barrier {:.0f} + reads+MFMA {:.0f} + barrier {:.0f}'
          ' + issue {:.0f}   (MFMA floor 512)'.format(step.mean(), wait_own.mean(), bar1.mean(), mfma.mean(), bar2.mean(),
                                                    issue.mean()))
    print('   per K step p10/p50/p90: wait {:.0f}/{:.0f}/{:.0f}  reads+MFMA {:.0f}/{:.0f}/{:.0f}  issue {:.0f}/{:.0f}/{:.0f}'.format(
        *np.percentile(wait_own, (10, 50, 90)), *np.percentile(mfma, (10, 50, 90)), *np.percentile(issue, (10, 50, 90))))
    # co-residency: workgroups per CU and their placement in time
    keys, counts = np.unique(cu_key, return_counts=True)
    print('   CUs used {}  workgroups per CU: min {} / mean {:.2f} / max {}   histogram {}'.format(
        len(keys), counts.min(), counts.mean(), counts.max(), dict(zip(*np.unique(counts, return_counts=True)))))
    rt0 = rt.min()
    span_ticks = (rt.max() - rt0)
    print('   workgroup start times: last start {:.1f} us after the first (kernel {:.1f} us)'.format(span_ticks / 100.0, t_traced))
    busiest = keys[np.argmax(counts)]
    sel = np.where(cu_key == busiest)[0]
    sel = sel[np.argsort(rt[sel])]
    print('   busiest CU (xcc {} se {} sh {} cu {}): start us / lifetime cycles / first-step wait / mean step'.format(
        busiest // 1000, (busiest // 100) % 10, (busiest % 100) // 16, busiest % 16))
    for i in sel[:8]:
        print('      wg {:5d} tile_m {:4d} tile_n {:2d}  start {:7.2f} us  life {:7d}  step {:6.0f}  simd {}'.format(
            i, tr[i, 9], tr[i, 10], (rt[i] - rt0) / 100.0, life[i], step[i].mean(), (hw[i] >> 4) & 3))
    sys.stdout.flush()
