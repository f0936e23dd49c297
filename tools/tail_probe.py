#!/usr/bin/env python
"""Does the last partial round of workgroups (launch tail) cost what the model says? Times the layer-3 shapes of
DeepLab v2 at pixel counts that give 500 / 512 / 526 / 768 / 1024 workgroups of the 128x128 tile."""
import sys
import os
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


print('{:<18s} {:>5s} {:>5s} {:>7s} {:>6s} {:>8s} {:>8s} {:>9s}'.format('shape', 'H', 'W', 'pixels', 'WGs', 'us', 'TF/s', 'us/Mpix'))
for name, cin, cout, k, dil in (('3x3d2 256->256', 256, 256, 3, 2), ('1x1 1024->256', 1024, 256, 1, 1),
                                ('1x1 256->1024', 256, 1024, 1, 1)):
    for (n, h, w) in ((20, 40, 40), (16, 32, 64), (20, 41, 41), (24, 32, 64), (30, 41, 41), (32, 32, 64), (40, 41, 41)):
        g = torch.Generator(device=DEV).manual_seed(0)
        pad = dil * (k - 1) // 2
        x = torch.randn(n, h, w, cin, generator=g, device=DEV).bfloat16()
        wp = (torch.randn(k * k, cout, cin, generator=g, device=DEV) * 0.05).bfloat16()
        scale = torch.ones(cout, device=DEV)
        bias = torch.zeros(cout, device=DEV)
        taps = ops.conv_taps(k, k, dil, pad)
        out = torch.empty(n, h, w, cout, dtype=torch.bfloat16, device=DEV)
        t = timeit(lambda: ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, relu=True, out=out))
        t1 = timeit(lambda: ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, relu=True, out=out, variant=4))
        pix = n * h * w
        wgs = ((pix + 127) // 128) * (cout // 128)
        print('{:<18s} {:5d} {:5d} {:7d} {:6d} {:8.1f} {:8.1f} {:9.1f}   2-stage {:8.1f}'.format(
            name, h, w, pix, wgs, t, 2.0 * pix * cin * cout * k * k / t / 1e6, t / pix * 1e6, t1))
