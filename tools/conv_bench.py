#!/usr/bin/env python
"""Micro-benchmark of csrc/conv.hip on the DeepLab v2 layer shapes (batch 20 = fused [sup; mix] pass at cfg 2)
against the library convolution (torch/MIOpen, bf16 channels-last). Prints one line per shape."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from cutmix_semisup_seg_amd import ops

DEV = 'cuda:0'
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [
    # name, H, W, Cin, Cout, k, stride, dil, count per forward
    ('l1 1x1 64->64', 81, 81, 64, 64, 1, 1, 1, 1), ('l1 3x3 64->64', 81, 81, 64, 64, 3, 1, 1, 3),
    ('l1 1x1 64->256', 81, 81, 64, 256, 1, 1, 1, 4), ('l1 1x1 256->64', 81, 81, 256, 64, 1, 1, 1, 2),
    ('l2 1x1s2 256->128', 81, 81, 256, 128, 1, 2, 1, 1), ('l2 1x1s2 256->512', 81, 81, 256, 512, 1, 2, 1, 1),
    ('l2 3x3 128->128', 41, 41, 128, 128, 3, 1, 1, 4), ('l2 1x1 128->512', 41, 41, 128, 512, 1, 1, 1, 4),
    ('l2 1x1 512->128', 41, 41, 512, 128, 1, 1, 1, 3),
    ('l3 1x1 512->256', 41, 41, 512, 256, 1, 1, 1, 1), ('l3 1x1 512->1024', 41, 41, 512, 1024, 1, 1, 1, 1),
    ('l3 1x1 1024->256', 41, 41, 1024, 256, 1, 1, 1, 22), ('l3 3x3d2 256->256', 41, 41, 256, 256, 3, 1, 2, 23),
    ('l3 1x1 256->1024', 41, 41, 256, 1024, 1, 1, 1, 23),
    ('l4 1x1 1024->512', 41, 41, 1024, 512, 1, 1, 1, 1), ('l4 1x1 1024->2048', 41, 41, 1024, 2048, 1, 1, 1, 1),
    ('l4 1x1 2048->512', 41, 41, 2048, 512, 1, 1, 1, 2), ('l4 3x3d4 512->512', 41, 41, 512, 512, 3, 1, 4, 3),
    ('l4 1x1 512->2048', 41, 41, 512, 2048, 1, 1, 1, 3),
]


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
    return e0.elapsed_time(e1) / iters


tot_mine = tot_lib = 0.0
print('{:<22s} {:>9s} {:>9s} {:>9s} {:>8s} {:>8s} {:>8s}'.format('shape', 'glds_us', 'tile64_us', 'lib_us', 'TF/s', 'GB/s', 'speedup'))
for name, H, W, Cin, Cout, k, stride, dil, cnt in SHAPES:
    g = torch.Generator(device=DEV).manual_seed(0)
    pad = dil * (k - 1) // 2
    x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
    w = (torch.randn(Cout, Cin, k, k, generator=g, device=DEV) * 0.05).bfloat16()
    wp = w.permute(2, 3, 0, 1).reshape(k * k, Cout, Cin).contiguous()
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    scale = torch.ones(Cout, device=DEV)
    bias = torch.zeros(Cout, device=DEV)
    taps = ops.conv_taps(k, k, dil, pad)
    out = torch.empty(N, Ho, Wo, Cout, dtype=torch.bfloat16, device=DEV)
    t_m = timeit(lambda: ops.conv_igemm(x, wp, taps, stride=stride, out_hw=(Ho, Wo), scale=scale, bias=bias, relu=True,
                                        out=out))
    t_r = timeit(lambda: ops.conv_igemm(x, wp, taps, stride=stride, out_hw=(Ho, Wo), scale=scale, bias=bias, relu=True,
                                        out=out, tile=64)) if Cout % 64 == 0 else float('nan')
    xcl = x.permute(0, 3, 1, 2)          # NCHW view with channels-last strides
    wcl = w.contiguous(memory_format=torch.channels_last)
    t_l = timeit(lambda: F.conv2d(xcl, wcl, None, stride, pad, dil))
    flops = 2.0 * N * Ho * Wo * Cout * Cin * k * k
    byts = 2.0 * (N * H * W * Cin / (stride * stride if k == 1 else 1) + N * Ho * Wo * Cout + k * k * Cin * Cout)
    print('{:<22s} {:9.1f} {:9.1f} {:9.1f} {:8.1f} {:8.0f} {:8.2f}'.format(name, t_m * 1e3, t_r * 1e3, t_l * 1e3, flops / t_m / 1e9,
                                                                  byts / t_m / 1e6, t_l / t_m))
    tot_mine += cnt * t_m
    tot_lib += cnt * t_l
print('weighted forward total: mine {:.2f} ms, library {:.2f} ms (conv only; library needs BN/ReLU/add passes on top)'.format(
    tot_mine, tot_lib))
