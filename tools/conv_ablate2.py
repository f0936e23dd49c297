#!/usr/bin/env python
"""Loads-only / MFMA-only ablation of conv_igemm per tile shape (variant 2: no MFMA phase, 3: no loads after the first
stage). If the kernel were bound by the L2 -> LDS fabric, the loads-only time of the 128 x 256 tile (0.75x the bytes
per FLOP) would be 0.75x that of the 128 x 128 tile."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import ops
DEV = 'cuda:0'


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


print('{:<24s} {:>6s} {:>9s} {:>9s} {:>9s}   {:>10s} {:>10s}'.format('shape', 'tile', 'full', 'no_mfma', 'no_loads', 'GB_to_LDS', 'TB/s loads'))
for name, N, H, W, Cin, Cout, k, dil in [('l4 3x3d4 512->512 n20', 20, 41, 41, 512, 512, 3, 4), ('l4 3x3d4 512->512 n40', 40, 41, 41, 512, 512, 3, 4),
                                         ('l3 1x1 1024->256 n40', 40, 41, 41, 1024, 256, 1, 1)]:
    g = torch.Generator(device=DEV).manual_seed(0)
    pad = dil * (k - 1) // 2
    x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
    wp = (torch.randn(k * k, Cout, Cin, generator=g, device=DEV) * 0.05).bfloat16()
    out = torch.empty(N, H, W, Cout, dtype=torch.bfloat16, device=DEV)
    taps = ops.conv_taps(k, k, dil, pad)
    M = N * H * W
    for tile, bm, bn in ((0, 128, 128), (256, 256, 128), (2256, 256, 128), (64, 128, 64)):
        ts = [timeit(lambda v=v: ops.conv_igemm(x, wp, taps, out=out, tile=tile, variant=v)) for v in (0, 2, 3)]
        wgs = ((M + bm - 1) // bm) * (Cout // bn)
        gb = wgs * (bm + bn) * 128 * (k * k * Cin // 64) / 1e9
        print('{:<24s} {:6d} {:9.1f} {:9.1f} {:9.1f}   {:10.2f} {:10.1f}'.format(name, tile, *ts, gb, gb / ts[1] * 1e3))
        if tile == 0:
            t2 = [timeit(lambda v=v: ops.conv_igemm(x, wp, taps, out=out, tile=tile, variant=v)) for v in (4, 6, 7)]
            print('{:<24s} {:>6s} {:9.1f} {:9.1f} {:9.1f}'.format('   two-stage (2 WG/CU)', '', *t2))
