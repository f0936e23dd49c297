#!/usr/bin/env python
"""Micro-benchmark of conv_wgrad on the DeepLab v2 layer shapes (batch 20), sweeping the split-K factor."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import ops
DEV = 'cuda:0'
N = 20
SHAPES = [('l1 3x3 64->64', 81, 81, 64, 64, 3, 1, 1, 3), ('l1 1x1 64->256', 81, 81, 64, 256, 1, 1, 1, 4),
          ('l1 1x1 256->64', 81, 81, 256, 64, 1, 1, 1, 2), ('l2 3x3 128->128', 41, 41, 128, 128, 3, 1, 1, 4),
          ('l2 1x1 128->512', 41, 41, 128, 512, 1, 1, 1, 4), ('l3 1x1 1024->256', 41, 41, 1024, 256, 1, 1, 1, 22),
          ('l3 3x3d2 256->256', 41, 41, 256, 256, 3, 1, 2, 23), ('l3 1x1 256->1024', 41, 41, 256, 1024, 1, 1, 1, 23),
          ('l4 1x1 2048->512', 41, 41, 2048, 512, 1, 1, 1, 2), ('l4 3x3d4 512->512', 41, 41, 512, 512, 3, 1, 4, 3),
          ('l4 1x1 512->2048', 41, 41, 512, 2048, 1, 1, 1, 3), ('l4 1x1 1024->2048', 41, 41, 1024, 2048, 1, 1, 1, 1)]


def timeit(fn, iters=10):
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


print('{:<20s} {:>8s} {:>8s} {:>8s} {:>8s} {:>8s} {:>8s} {:>8s}'.format('shape', 'auto', 'ks=2', 'ks=4', 'ks=8', 'ks=16', 'TF/s', ''))
tot = 0.0
for name, H, W, Cin, Cout, k, stride, dil, cnt in SHAPES:
    g = torch.Generator(device=DEV).manual_seed(0)
    pad = dil * (k - 1) // 2
    x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
    du = torch.randn(N, H, W, Cout, generator=g, device=DEV).bfloat16()
    dw = torch.zeros(k * k, Cout, Cin, device=DEV)
    taps = ops.conv_taps(k, k, dil, pad)
    ts = [timeit(lambda ks=ks: ops.conv_wgrad(du, x, taps, dw, ksplit=ks)) for ks in (0, 2, 4, 8)]
    ts.append(timeit(lambda: ops.conv_wgrad(du, x, taps, dw, ksplit=16)))
    flops = 2.0 * N * H * W * Cout * Cin * k * k
    print('{:<20s} {:8.1f} {:8.1f} {:8.1f} {:8.1f} {:8.1f} {:8.1f}'.format(name, *ts, flops / min(ts) / 1e6))
    tot += cnt * ts[0]
print('weighted total (auto): {:.2f} ms'.format(tot / 1e3))
