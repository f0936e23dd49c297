#!/usr/bin/env python
"""The eight-phase 256 x 256 weight gradient (csrc/wgrad8.hip) against the 128 x 128 kernel (csrc/conv.hip) and an fp32
matrix product of the same bf16 operands: odd geometries (taps, strides, partial last K tile, slices of odd length), repeated
launches; then the timing of the DeepLab v2 layer shapes of BASELINE configs[1] / configs[2], alone and as the three launches
of a bottleneck on three streams, and a cycle trace (where a workgroup's cycles go).
    python tools/wgrad8_check.py [check] [time] [trace]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cutmix_semisup_seg_amd import ops
from cutmix_semisup_seg_amd._lib import lib

DEV = 'cuda:0'
PERF = [
    # name, N, H, W, Cin, Cout, k, dil
    ('c2 l3 1x1 1024->256', 20, 41, 41, 1024, 256, 1, 1),
    ('c2 l3 3x3d2 256->256', 20, 41, 41, 256, 256, 3, 2),
    ('c2 l3 1x1 256->1024', 20, 41, 41, 256, 1024, 1, 1),
    ('c2 l4 1x1 2048->512', 20, 41, 41, 2048, 512, 1, 1),
    ('c2 l4 3x3d4 512->512', 20, 41, 41, 512, 512, 3, 4),
    ('c2 l4 1x1 512->2048', 20, 41, 41, 512, 2048, 1, 1),
    ('c3 l3 1x1 1024->256', 8, 65, 129, 1024, 256, 1, 1),
    ('c3 l3 3x3d2 256->256', 8, 65, 129, 256, 256, 3, 2),
    ('c3 l3 1x1 256->1024', 8, 65, 129, 256, 1024, 1, 1),
    ('c3 l4 3x3d4 512->512', 8, 65, 129, 512, 512, 3, 4),
]
SMALL = [
    # name, N, H, W, Cin, Cout, k, dil, stride
    ('1x1, partial last K tile', 3, 19, 21, 256, 256, 1, 1, 1),
    ('3x3 dilation 2', 2, 23, 37, 256, 256, 3, 2, 1),
    ('3x3 stride 2', 4, 33, 47, 256, 512, 3, 1, 2),
    ('1x1 stride 2 (shortcut)', 3, 41, 41, 512, 256, 1, 1, 2),
    ('3x3 dilation 4, two ci tiles', 1, 41, 41, 512, 256, 3, 4, 1),
    ('narrow map (Wo = 9)', 4, 40, 9, 256, 256, 3, 1, 1),
]


def make(case, seed=0):
    name, N, H, W, Cin, Cout, k, dil = case[:8]
    stride = case[8] if len(case) > 8 else 1
    g = torch.Generator(device=DEV).manual_seed(seed)
    pad = dil * (k - 1) // 2
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
    du = (torch.randn(N, Ho, Wo, Cout, generator=g, device=DEV) * 0.05).bfloat16()
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    return du, x, ops.conv_taps(k, k, dil, pad), scale, stride, (k, dil, pad)


def run(t, mode, dw=None, ksplit=0, wg_target=0):
    du, x, taps, scale, stride, _ = t
    if dw is None:
        dw = torch.zeros(len(taps), du.shape[3], x.shape[3], device=DEV)
    lib.cms_conv_set_wgrad8(mode)
    try:
        if mode == 1 and ksplit == 0 and wg_target == 0:
            wg_target = 256                      # (alone the dispatcher would send launches of few tiles to the 128 x 128 kernel)
        if mode == 1 and ops.conv_wgrad(du, x, taps, dw, stride=stride, scale=scale, ksplit=ksplit, query_kernel=True,
                                        wg_target=wg_target) != 8:
            raise RuntimeError('the eight-phase kernel does not take this launch')
        ops.conv_wgrad(du, x, taps, dw, stride=stride, scale=scale, ksplit=ksplit, wg_target=wg_target)
    finally:
        lib.cms_conv_set_wgrad8(-1)
    return dw


def reference(t):
    """fp32 weight gradient of the same bf16 operands (ATen)."""
    du, x, taps, scale, stride, (k, dil, pad) = t
    xf = x.float().permute(0, 3, 1, 2)
    gf = du.float().permute(0, 3, 1, 2)
    w = torch.nn.grad.conv2d_weight(xf, (du.shape[3], x.shape[3], k, k), gf, stride=stride, padding=pad, dilation=dil)
    return (w * scale.view(-1, 1, 1, 1)).permute(2, 3, 0, 1).reshape(k * k, du.shape[3], x.shape[3])


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def check():
    bad = 0
    for case in SMALL + PERF[:3]:
        t = make(case)
        ref = reference(t)
        old = run(t, 0)
        den = ref.abs().max().item()
        for ks in (0, 1, 3):
            errs = []
            for rep in range(3):
                new = run(t, 1, ksplit=ks)
                errs.append((new - ref).abs().max().item() / den)
            e_old = (old - ref).abs().max().item() / den
            ok = max(errs) <= 2e-5
            bad += 0 if ok else 1
            print('{:32s} ksplit {}: 8-phase vs fp32 {:.2e} (3 launches)   128x128 vs fp32 {:.2e}   {}'.format(
                case[0], ks, max(errs), e_old, 'ok' if ok else 'WRONG'))
    print('MISMATCHES:', bad)
    return bad


def time_shapes():
    print('shape                         128x128 us   8-phase us    PF/s old / new')
    for case in PERF:
        t = make(case)
        du, x, taps = t[0], t[1], t[2]
        dw = torch.zeros(len(taps), du.shape[3], x.shape[3], device=DEV)
        flops = 2.0 * du.shape[0] * du.shape[1] * du.shape[2] * du.shape[3] * x.shape[3] * len(taps)
        a = timeit(lambda: run(t, 0, dw))
        b = timeit(lambda: run(t, 1, dw))
        print('{:28s} {:10.1f} {:12.1f}      {:.2f} / {:.2f}'.format(case[0], a, b, flops / a * 1e-9, flops / b * 1e-9))
    # the three launches of a layer-3 bottleneck on three streams (what the weight-gradient side of the backward pass issues)
    for cfg in ('c2', 'c3'):
        ts = [make(c) for c in PERF if c[0].startswith(cfg + ' l3')]
        dws = [torch.zeros(len(t[2]), t[0].shape[3], t[1].shape[3], device=DEV) for t in ts]
        flops = sum(2.0 * t[0].numel() * t[1].shape[3] * len(t[2]) for t in ts)
        streams = [torch.cuda.Stream() for _ in ts]
        for mode in (0, 1):
            for nstreams in (1, 3):
                def body():
                    cur = torch.cuda.current_stream()
                    for i, (t, dw) in enumerate(zip(ts, dws)):
                        s = streams[i] if nstreams == 3 else cur
                        if nstreams == 3:
                            s.wait_stream(cur)
                        with torch.cuda.stream(s):
                            run(t, mode, dw)
                    if nstreams == 3:
                        for s in streams:
                            cur.wait_stream(s)
                us = timeit(body)
                print('{} layer-3 bottleneck, {} kernel, {} stream(s): {:.1f} us = {:.2f} PF/s'.format(
                    cfg, '8-phase' if mode else '128x128', nstreams, us, flops / us * 1e-9))


def trace():
    for case in PERF[:3] + PERF[6:8]:
        t = make(case)
        nwg = 1024
        buf = torch.zeros(nwg * 16, dtype=torch.int32, device=DEV)
        run(t, 1)
        torch.cuda.synchronize()
        lib.cms_conv_set_trace(buf.data_ptr(), nwg)
        run(t, 1)
        torch.cuda.synchronize()
        lib.cms_conv_set_trace(None, 0)
        tr = buf.cpu().numpy().view(np.uint32).reshape(nwg, 16).astype(np.int64)
        tr = tr[tr[:, 12] > 0]
        d = lambda a, b: ((tr[:, a] - tr[:, b]) & 0xffffffff)
        kt = tr[:, 12]
        print('== {}: {} workgroups, {:.0f} K tiles each'.format(case[0], len(tr), kt.mean()))
        print('   prologue {:.0f} | K loop {:.0f} = {:.0f} per K tile (MFMA floor 2048) | epilogue issue {:.0f} | acknowledged {:.0f}'.format(
            d(1, 0).mean(), d(2, 1).mean(), (d(2, 1) / (kt + (kt & 1))).mean(), d(3, 2).mean(), d(4, 3).mean()))


if __name__ == '__main__':
    what = sys.argv[1:] or ['check', 'time', 'trace']
    rc = 0
    if 'check' in what:
        rc = check()
    if 'time' in what:
        time_shapes()
    if 'trace' in what:
        trace()
    sys.exit(1 if rc else 0)
