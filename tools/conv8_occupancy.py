#!/usr/bin/env python
"""How the eight-phase convolution (csrc/conv8.hip) behaves as the machine fills: one layer shape at a growing number of
output tiles (batch rows), alone and as two streams launching the same layer concurrently. Reports per launch: wall time,
PF/s, the traced run length in cycles and the K-loop cycles per K tile -- so that a slow-down can be told apart as "more
cycles" (contention for L2 / HBM / the fabric) or "same cycles, longer wall time" (clock) or "a second round of tiles".
    python tools/conv8_occupancy.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cutmix_semisup_seg_amd import ops
from cutmix_semisup_seg_amd._lib import lib
from conv8_check import make, run

SHAPES = [('1x1 1024->256', 1024, 256, 1, 1), ('3x3d2 256->256', 256, 256, 3, 2)]
TILES = [64, 120, 128, 132, 192, 250, 256, 264]


def traced(t, epi):
    nwg = 2048
    buf = torch.zeros(nwg * 64, dtype=torch.int32, device='cuda:0')
    lib.cms_conv_set_trace(buf.data_ptr(), nwg)
    run(t, epi, 92)
    torch.cuda.synchronize()
    lib.cms_conv_set_trace(None, 0)
    tr = buf.cpu().numpy().view(np.uint32).reshape(nwg, 4, 16).astype(np.int64)
    runs = tr[tr[:, :, 12] > 0]
    d = lambda a, b: ((runs[:, a] - runs[:, b]) & 0xffffffff)
    return d(8, 0).mean(), (d(3, 2) / runs[:, 12]).mean(), len(runs)


def wall(fns, iters=30):
    """fns: one callable per stream; every stream launches `iters` times back to back; returns us per launch-round."""
    streams = [torch.cuda.Stream() for _ in fns]
    for s, f in zip(streams, fns):
        with torch.cuda.stream(s):
            for _ in range(3):
                f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    for _ in range(iters):
        for s, f in zip(streams, fns):
            with torch.cuda.stream(s):
                f()
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, cin, cout, k, dil in SHAPES:
    print('== {}   (rows: tiles of 256 pixels per launch)'.format(name))
    print('   tiles |  one stream: us   PF/s  run cycles  K-tile cycles | two streams: us per pair   PF/s   pair / single')
    for tiles in TILES:
        # N images of 16 x 16 pixels = one tile each
        case = (name, tiles, 16, 16, cin, cout, k, dil)
        ta, tb = make(case, 0), make(case, 1)
        oa, ob = run(ta, 'relu', 90), run(tb, 'relu', 90)
        flops = 2.0 * tiles * 256 * cin * k * k * cout
        one = wall([lambda: run(ta, 'relu', 90, out=oa)])
        two = wall([lambda: run(ta, 'relu', 90, out=oa), lambda: run(tb, 'relu', 90, out=ob)])
        cyc, ktc, nr = traced(ta, 'relu')
        print('   {:5d} | {:14.1f} {:6.2f} {:11.0f} {:14.0f} | {:20.1f} {:8.2f} {:12.2f}'.format(
            tiles, one, flops / one * 1e-9, cyc, ktc, two, 2 * flops / two * 1e-9, two / one))
