#!/usr/bin/env python
"""Same occupancy (16 waves / CU), loads overlapped with MFMAs: 8-wave workgroups on the 128 x 128 tile with two LDS
stages (tile 1128, variant 4) against the default (4 waves, one stage, 4 workgroups / CU)."""
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


g = torch.Generator(device=DEV).manual_seed(0)
print('{:<22s} {:>9s} {:>9s} {:>9s} {:>9s} {:>9s}'.format('shape', 'default', '8w_1stg', '8w_2stg', '8w256_2s', '4w_2stg'))
for name, n, cin, cout, k, dil in (('l3 1x1 1024->256', 20, 1024, 256, 1, 1), ('l3 3x3d2 256->256', 20, 256, 256, 3, 2),
                                   ('l3 1x1 256->1024', 20, 256, 1024, 1, 1), ('l4 3x3d4 512->512', 20, 512, 512, 3, 4),
                                   ('l4 1x1 2048->512', 20, 2048, 512, 1, 1), ('l4 3x3d4 n40', 40, 512, 512, 3, 4)):
    pad = dil * (k - 1) // 2
    x = torch.randn(n, 41, 41, cin, generator=g, device=DEV).bfloat16()
    wp = (torch.randn(k * k, cout, cin, generator=g, device=DEV) * 0.05).bfloat16()
    scale, bias = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    taps = ops.conv_taps(k, k, dil, pad)
    out = torch.empty(n, 41, 41, cout, dtype=torch.bfloat16, device=DEV)
    ts = [timeit(lambda t=t, v=v: ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, relu=True, out=out, tile=t, variant=v))
          for t, v in ((0, 0), (1128, 0), (1128, 4), (256, 4), (0, 4))]
    print('{:<22s} '.format(name) + ' '.join('{:9.1f}'.format(t) for t in ts))
