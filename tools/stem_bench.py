#!/usr/bin/env python
"""Times the stem kernels (forward, weight gradient, pooling) at the BASELINE configs[1] geometry (20 x 3 x 321 x 321)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import ops
DEV = 'cuda:0'
g = torch.Generator(device=DEV).manual_seed(0)
N, H, W = 20, 321, 321
x = torch.randn(N, 3, H, W, generator=g, device=DEV).bfloat16()
w = (torch.randn(49, 64, 3, generator=g, device=DEV) * 0.1)
scale, bias = torch.rand(64, generator=g, device=DEV) + 0.5, torch.randn(64, generator=g, device=DEV) * 0.1
w147 = ops.stem_pack_weights(w)


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


s = ops.stem_forward(x, w147, scale, bias, torch.bfloat16)
p, idx = ops.maxpool3x3s2_forward(s)
dp = torch.randn(p.shape, generator=g, device=DEV).bfloat16()
ds = ops.maxpool3x3s2_relu_backward(dp, idx, s)
dw = torch.zeros(49, 64, 3, device=DEV)
print('stem_forward        {:8.1f} us'.format(timeit(lambda: ops.stem_forward(x, w147, scale, bias, torch.bfloat16))))
print('maxpool_forward     {:8.1f} us'.format(timeit(lambda: ops.maxpool3x3s2_forward(s))))
print('maxpool_relu_bwd    {:8.1f} us'.format(timeit(lambda: ops.maxpool3x3s2_relu_backward(dp, idx, s))))
print('stem_wgrad          {:8.1f} us'.format(timeit(lambda: ops.stem_wgrad(x, ds, dw, scale))))
