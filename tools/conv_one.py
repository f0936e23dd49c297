#!/usr/bin/env python
"""Run one convolution shape a few times (for rocprofv3 --pmc passes). usage: conv_one.py H W Cin Cout k dil [N] [tile]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import ops
H, W, Cin, Cout, k, dil = (int(v) for v in sys.argv[1:7])
N = int(sys.argv[7]) if len(sys.argv) > 7 else 20
tile = int(sys.argv[8]) if len(sys.argv) > 8 else 0
DEV = 'cuda:0'
g = torch.Generator(device=DEV).manual_seed(0)
pad = dil * (k - 1) // 2
x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
wp = (torch.randn(k * k, Cout, Cin, generator=g, device=DEV) * 0.05).bfloat16()
scale, bias = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
out = torch.empty(N, H, W, Cout, dtype=torch.bfloat16, device=DEV)
for _ in range(5):
    ops.conv_igemm(x, wp, ops.conv_taps(k, k, dil, pad), scale=scale, bias=bias, relu=True, out=out, tile=tile)
torch.cuda.synchronize()
