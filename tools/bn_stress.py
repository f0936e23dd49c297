"""Stress test of the last-block protocol of bn_reduce_tiled_kernel (csrc/bn.hip): partial sums travel through write-through
stores + an explicit s_waitcnt, a counter atomic, and sc1 loads in the last block -- no fences. Thousands of launches alternate
between two inputs on ONE workspace while another stream keeps the L2s busy with convolutions; every result must equal,
bit for bit, the first result computed for that input (a stale partial would be the other input's).
CMS_BN_FENCE=1 python tools/bn_stress.py runs the fenced (release / acquire) variant of the same kernel."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import ops

DEV = 'cuda:0'
torch.manual_seed(0)
bad = total = 0
side = torch.cuda.Stream()
xc = torch.randn(20, 41, 41, 256, device=DEV).bfloat16()
wc = (torch.randn(9, 256, 256, device=DEV) * 0.05).bfloat16()
taps = ops.conv_taps(3, 3, 2, 2)
for (P, C, G) in [(16810, 1024, 1), (33620, 256, 2), (65610, 64, 1), (16810, 2048, 2), (4225 * 4, 2208, 1)]:
    xs = [(torch.randn(P, C, device=DEV) * (1.0 + i) + i).bfloat16() for i in range(2)]
    dy = torch.randn(P, C, device=DEV).bfloat16()
    y = torch.relu(torch.randn(P, C, device=DEV)).bfloat16()
    ws = ops.bn_workspace(P, C, DEV, G)
    mean, rstd = torch.zeros(G * C, device=DEV), torch.ones(G * C, device=DEV)
    ref = {}
    outs = []
    for it in range(1500):
        i, mode = it & 1, (it >> 1) & 1
        if it % 50 == 0:
            with torch.cuda.stream(side):
                for _ in range(8):
                    ops.conv_igemm(xc, wc, taps)
        out = torch.empty(G * 2 * C, dtype=torch.float64, device=DEV)
        if mode == 0:
            ops.bn_op('reduce', c=C, dtype=torch.bfloat16, n_pixels=P, groups=G, x=xs[i], sums=out, ws=ws)
        else:
            ops.bn_op('reduce_bwd', c=C, dtype=torch.bfloat16, n_pixels=P, groups=G, x=xs[i], dy=dy, y=y, mean=mean, rstd=rstd,
                      sums=out, ws=ws)
        outs.append(((i, mode), out))
        if len(outs) == 100:
            for key, o in outs:
                r = ref.setdefault(key, o)
                total += 1
                if not torch.equal(r, o):
                    bad += 1
            outs = []
    torch.cuda.synchronize()
    print('P=%d C=%d groups=%d: %d launches checked, %d mismatches so far' % (P, C, G, total, bad))
print('bn_stress', 'OK' if bad == 0 else 'FAILED', total, bad)
sys.exit(1 if bad else 0)
