"""Micro-benchmark of the batch-statistics BatchNorm kernels (csrc/bn.hip) alone on one stream, on the unit shapes of the
DeepLab v2 step at cfg 2 (student batch 10): per launch time and the HBM rate of the bytes the launch has to move.
    python tools/bn_bench.py [--dtype bf16|fp32] [--reps 50]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutmix_semisup_seg_amd import ops  # noqa: E402

SHAPES = [(65610, 64), (65610, 256), (16810, 128), (16810, 512), (16810, 256), (16810, 1024), (16810, 2048)]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3       # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--reps', type=int, default=50)
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    es = 2 if dt == torch.bfloat16 else 4
    dev = 'cuda:0'
    print('%-14s %10s %10s %10s %10s %10s %10s %10s   (us; GB/s of compulsory bytes)' % (
        'pixels x C', 'reduce', 'reduce_ws', 'stats', 'apply', 'red_bwd', 'red_bwd_ws', 'bwd_apply'))
    for P, C in SHAPES:
        x = torch.randn(P, C, device=dev).to(dt)
        dy = torch.randn(P, C, device=dev).to(dt)
        y = torch.relu(torch.randn(P, C, device=dev)).to(dt)
        out, dres = torch.empty_like(x), torch.empty_like(x)
        sums = torch.zeros(2 * C, dtype=torch.float64, device=dev)
        ws = ops.bn_workspace(P, C, dev)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        m, r, sc, sh = (torch.zeros(C, device=dev) for _ in range(4))
        r.fill_(1.0)
        sc.fill_(1.0)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        kw = dict(c=C, dtype=dt, n_pixels=P)
        t = {}
        t['reduce'] = timeit(lambda: ops.bn_op('reduce', x=x, sums=sums, **kw), a.reps)
        t['reduce_ws'] = timeit(lambda: ops.bn_op('reduce', x=x, sums=sums, ws=ws, **kw), a.reps)
        t['stats'] = timeit(lambda: ops.bn_op('stats', x=x, ws=ws, gamma=g, beta=b, mean=m, rstd=r, scale=sc, shift=sh,
                                              running_mean=rm, running_var=rv, **kw), a.reps)
        m.zero_(); r.fill_(1.0); sc.fill_(1.0); sh.zero_()
        t['apply'] = timeit(lambda: ops.bn_op('apply', x=x, y=out, scale=sc, shift=sh, relu=True, **kw), a.reps)
        t['red_bwd'] = timeit(lambda: ops.bn_op('reduce_bwd', x=x, dy=dy, y=y, mean=m, rstd=r, sums=sums, **kw), a.reps)
        t['red_bwd_ws'] = timeit(lambda: ops.bn_op('reduce_bwd', x=x, dy=dy, y=y, mean=m, rstd=r, sums=sums, ws=ws, **kw), a.reps)
        t['bwd_apply'] = timeit(lambda: ops.bn_op('bwd_apply', count=P, x=x, dy=dy, y=y, dx=out, dres=None, mean=m, rstd=r,
                                                  gamma=g, sums=sums, **kw), a.reps)
        nb = P * C * es
        byt = dict(reduce=nb, reduce_ws=nb, stats=nb, apply=2 * nb, red_bwd=3 * nb, red_bwd_ws=3 * nb, bwd_apply=4 * nb)
        print('%-14s ' % ('%dx%d' % (P, C)) + ' '.join('%5.1f/%4.0f' % (t[k], byt[k] / t[k] * 1e-3) for k in
                                                       ('reduce', 'reduce_ws', 'stats', 'apply', 'red_bwd', 'red_bwd_ws', 'bwd_apply')))


if __name__ == '__main__':
    main()
