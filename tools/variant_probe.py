#!/usr/bin/env python
"""Loader variants of conv_igemm on the DeepLab v2 layer shapes: 0 = one 64-wide stage, 5 = two 32-wide stages."""
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


VARIANTS = [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['0', '5'])]
g = torch.Generator(device=DEV).manual_seed(0)
tot = [0.0] * len(VARIANTS)
for name, n, hw, cin, cout, k, dil, cnt in (('l1 3x3 64->64', 20, 81, 64, 64, 3, 1, 3), ('l1 1x1 64->256', 20, 81, 64, 256, 1, 1, 4),
                                            ('l2 3x3 128->128', 20, 41, 128, 128, 3, 1, 4), ('l2 1x1 128->512', 20, 41, 128, 512, 1, 1, 4),
                                            ('l3 1x1 1024->256', 20, 41, 1024, 256, 1, 1, 22), ('l3 3x3d2 256->256', 20, 41, 256, 256, 3, 2, 23),
                                            ('l3 1x1 256->1024', 20, 41, 256, 1024, 1, 1, 23), ('l4 1x1 2048->512', 20, 41, 2048, 512, 1, 1, 2),
                                            ('l4 3x3d4 512->512', 20, 41, 512, 512, 3, 4, 3), ('l4 1x1 512->2048', 20, 41, 512, 2048, 1, 1, 3),
                                            ('l3 3x3d2 n40', 40, 41, 256, 256, 3, 2, 0)):
    pad = dil * (k - 1) // 2
    x = torch.randn(n, hw, hw, cin, generator=g, device=DEV).bfloat16()
    wp = (torch.randn(k * k, cout, cin, generator=g, device=DEV) * 0.05).bfloat16()
    scale, bias = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    taps = ops.conv_taps(k, k, dil, pad)
    out = torch.empty(n, hw, hw, cout, dtype=torch.bfloat16, device=DEV)
    ts = [timeit(lambda v=v: ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, relu=True, out=out, variant=v)) for v in VARIANTS]
    for i, t in enumerate(ts):
        tot[i] += cnt * t
    print('{:<20s}'.format(name), ' '.join('{:8.1f}'.format(t) for t in ts), ' TF/s best {:.0f}'.format(
        2.0 * n * hw * hw * cin * cout * k * k / min(ts) / 1e6))
print('weighted forward totals (ms):', ' '.join('{:.2f}'.format(t / 1e3) for t in tot), 'variants', VARIANTS)
