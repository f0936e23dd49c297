#!/usr/bin/env python
"""Variant sweep of the direct-to-LDS MFMA convolution (csrc/conv.hip) on representative DeepLab v2 layer shapes of
BASELINE configs[1] (fused batch 20, 41 x 41) and configs[2] (fused batch 8, 65 x 129): default kernel vs the pipelined
variants 10..14 on the 128 x 128 / 4-wave and 128 x 256 / 8-wave tiles. Every variant is first checked against the
default kernel's output (three launches each: a stage-ring race shows up as a mismatch that comes and goes)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import ops

DEV = 'cuda:0'
SHAPES = [
    # name, N, H, W, Cin, Cout, k, dil, residual
    ('c2 l3 1x1 1024->256', 20, 41, 41, 1024, 256, 1, 1, False),
    ('c2 l3 3x3d2 256->256', 20, 41, 41, 256, 256, 3, 2, False),
    ('c2 l3 1x1 256->1024+res', 20, 41, 41, 256, 1024, 1, 1, True),
    ('c2 l4 3x3d4 512->512', 20, 41, 41, 512, 512, 3, 4, False),
    ('c2 l4 1x1 512->2048+res', 20, 41, 41, 512, 2048, 1, 1, True),
    ('c2 l2 1x1 128->512+res', 20, 41, 41, 128, 512, 1, 1, True),
    ('c2 l1 1x1 64->256+res', 20, 81, 81, 64, 256, 1, 1, True),
    ('c3 l3 1x1 1024->256', 8, 65, 129, 1024, 256, 1, 1, False),
    ('c3 l3 3x3d2 256->256', 8, 65, 129, 256, 256, 3, 2, False),
    ('c3 l3 1x1 256->1024+res', 8, 65, 129, 256, 1024, 1, 1, True),
    ('c3 l4 3x3d4 512->512', 8, 65, 129, 512, 512, 3, 4, False),
]
VARIANTS = [(0, 0)] + [(0, v) for v in (10, 11, 12, 13, 14)] + [(256, 0)] + [(256, v) for v in (10, 11, 12, 13, 14)]
if os.environ.get('CMS_VARIANTS'):            # e.g. CMS_VARIANTS=0:0,0:20,0:21 (tile:variant)
    VARIANTS = [tuple(int(v) for v in p.split(':')) for p in os.environ['CMS_VARIANTS'].split(',')]
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if any(a in s[0] for a in sys.argv[1:])]


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
    return e0.elapsed_time(e1) / iters * 1e3       # us


print('{:<26s}'.format('shape') + ''.join('{:>10s}'.format('t{}v{}'.format(t, v)) for t, v in VARIANTS) + '   best   PF/s')
bad = []
for name, N, H, W, Cin, Cout, k, dil, use_res in SHAPES:
    g = torch.Generator(device=DEV).manual_seed(0)
    pad = dil * (k - 1) // 2
    x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
    wp = (torch.randn(k * k, Cout, Cin, generator=g, device=DEV) * (2.0 / (Cin * k * k)) ** 0.5).bfloat16()
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    bias = torch.randn(Cout, generator=g, device=DEV) * 0.1
    res = torch.randn(N, H, W, Cout, generator=g, device=DEV).bfloat16() if use_res else None
    taps = ops.conv_taps(k, k, dil, pad)
    ref = ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, res=res, relu=True)
    flops = 2.0 * N * H * W * Cout * Cin * k * k
    row, times = '{:<26s}'.format(name), []
    for tile, var in VARIANTS:
        out = torch.empty_like(ref)
        ok = True
        for _ in range(3):
            out.fill_(7.0)
            ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, res=res, relu=True, out=out, tile=tile, variant=var)
            if var >= 20:      # K rotation changes the summation order: one bf16 ulp of the output
                if not torch.allclose(out.float(), ref.float(), rtol=2 ** -7, atol=2e-2):
                    ok = False
            elif not torch.equal(out, ref):
                ok = False
        if not ok:
            bad.append((name, tile, var, float((out.float() - ref.float()).abs().max())))
        t = timeit(lambda: ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, res=res, relu=True, out=out, tile=tile,
                                          variant=var))
        times.append(t)
        row += '{:>10s}'.format('{:.1f}{}'.format(t, '' if ok else '!'))
    best = min(range(len(times)), key=lambda i: times[i])
    print(row + '   t{}v{}  {:.2f}'.format(VARIANTS[best][0], VARIANTS[best][1], flops / times[best] / 1e9), flush=True)
print('MISMATCHES vs the default kernel:', bad if bad else 'none')
