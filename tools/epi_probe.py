#!/usr/bin/env python
"""What do the residual / ReLU-mask reads of the convolution epilogue cost? Layer-3 shapes of DeepLab v2, batch 20."""
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


N, H, W = 20, 41, 41
g = torch.Generator(device=DEV).manual_seed(0)
for name, cin, cout, k, dil in (('1x1 256->1024', 256, 1024, 1, 1), ('1x1 1024->256', 1024, 256, 1, 1),
                                ('3x3d2 256->256', 256, 256, 3, 2), ('1x1 512->2048', 512, 2048, 1, 1)):
    pad = dil * (k - 1) // 2
    x = torch.randn(N, H, W, cin, generator=g, device=DEV).bfloat16()
    wp = (torch.randn(k * k, cout, cin, generator=g, device=DEV) * 0.05).bfloat16()
    scale, bias = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    taps = ops.conv_taps(k, k, dil, pad)
    out = torch.empty(N, H, W, cout, dtype=torch.bfloat16, device=DEV)
    res = torch.randn(N, H, W, cout, generator=g, device=DEV).bfloat16()
    msk = torch.randn(N, H, W, cout, generator=g, device=DEV).bfloat16()
    t0 = timeit(lambda: ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, relu=True, out=out))
    t1 = timeit(lambda: ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, relu=True, res=res, out=out))
    t2 = timeit(lambda: ops.conv_igemm(x, wp, taps, mode=1, out=out))
    t3 = timeit(lambda: ops.conv_igemm(x, wp, taps, mode=1, mask_src=msk, out=out))
    t4 = timeit(lambda: ops.conv_igemm(x, wp, taps, mode=1, mask_src=msk, res=res, out=out))
    print('{:<16s} fwd {:6.1f}  fwd+res {:6.1f}  | dgrad plain {:6.1f}  +mask {:6.1f}  +mask+res {:6.1f}  us'.format(
        name, t0, t1, t2, t3, t4))
