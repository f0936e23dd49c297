#!/usr/bin/env python
"""The ASPP head's forward pieces alone (csrc/conv.hip fp32-NCHW epilogue, csrc/aspp.hip gather), BASELINE configs[1] / [2]
geometry: Z[n][tap*C + c] = <W[tap][c], x> as ONE 1x1 GEMM (M pixels x 384 rows x K = 2048) + the gather of the 18 shifted planes.
For comparison the same GEMM with a bf16 NHWC output on the 128 x 128 kernel and -- rows padded to 512 -- on the eight-phase kernel:
how much of the head's launch time is the fp32 NCHW epilogue, how much the tile shape."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import ops

DEV = 'cuda:0'


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, N, h, w, C in (('cfg2 20x41x41 C=21', 20, 41, 41, 21), ('cfg3 8x65x129 C=19', 8, 65, 129, 19)):
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(N, h, w, 2048, generator=g, device=DEV).bfloat16()
    zc = (18 * C + 127) // 128 * 128
    wall = (torch.randn(1, zc, 2048, generator=g, device=DEV) * 0.02).bfloat16()
    wall512 = torch.zeros(1, 512, 2048, dtype=torch.bfloat16, device=DEV)
    wall512[:, :zc] = wall
    z = torch.empty(N, zc, h, w, dtype=torch.float32, device=DEV)
    bias = torch.zeros(C, device=DEV)
    taps = ops.conv_taps(3, 3, 6, 6) + ops.conv_taps(3, 3, 12, 12)
    one = [(0, 0)]
    y = torch.empty(N, h, w, zc, dtype=torch.bfloat16, device=DEV)
    y512 = torch.empty(N, h, w, 512, dtype=torch.bfloat16, device=DEV)
    gf = 2.0 * N * h * w * zc * 2048
    t_z = timed(lambda: ops.conv_igemm(x, wall, one, out_f32_nchw=z, cout_real=zc))
    t_g = timed(lambda: ops.aspp_gather_fwd(z, bias, taps, C))
    t_b = timed(lambda: ops.conv_igemm(x, wall, one, out=y, variant=99))
    t_8 = timed(lambda: ops.conv_igemm(x, wall512, one, out=y512, variant=90))
    t_88 = timed(lambda: ops.conv_igemm(x, wall512, one, out=y512, variant=99))
    print('{}: Z GEMM fp32 NCHW {:.1f} us ({:.2f} PF/s)  gather {:.1f} us | same GEMM, bf16 NHWC out: 128x128 kernel {:.1f} us, '
          'rows padded to 512: eight-phase {:.1f} us / 128x128 {:.1f} us'.format(name, t_z, gf / t_z / 1e9, t_g, t_b, t_8, t_88))
