#!/usr/bin/env python
"""Cycle trace of the eight-phase convolution (csrc/conv8.hip, variants 92 = whole tiles / 93 = stream-K with stamps):
where the cycles of a run go -- tables, first loads, K loop (per K tile), partial-tile hand-off, epilogue.
    python tools/conv8_trace.py [shape-name substrings ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cutmix_semisup_seg_amd import ops
from cutmix_semisup_seg_amd._lib import lib
from conv8_check import PERF, make, run, timeit

names = [a for a in sys.argv[1:]]
cases = [c for c in PERF if not names or any(s in c[0] for s in names)]
for case in cases:
    t = make(case)
    epi = 'relu' if case[5] < 1024 else 'res_relu'
    out = run(t, epi, 99)
    for var in (92, 93):
        nwg = 2048
        buf = torch.zeros(nwg * 64, dtype=torch.int32, device='cuda:0')
        for _ in range(2):
            run(t, epi, var - 2, out=out)
        torch.cuda.synchronize()
        t_plain = timeit(lambda: run(t, epi, var - 2, out=out), 10)
        lib.cms_conv_set_trace(buf.data_ptr(), nwg)
        run(t, epi, var, out=out)
        torch.cuda.synchronize()
        lib.cms_conv_set_trace(None, 0)
        tr = buf.cpu().numpy().view(np.uint32).reshape(nwg, 4, 16).astype(np.int64)
        runs = tr[tr[:, :, 12] > 0]                      # (n_runs, 16)
        if len(runs) == 0:
            print(case[0], var, 'no runs traced')
            continue
        d = lambda a, b: ((runs[:, a] - runs[:, b]) & 0xffffffff)
        kt = runs[:, 12]
        partial = (runs[:, 14] & 2) != 0
        last = (runs[:, 14] & 1) != 0
        full = ~partial
        print('== {} v{}: {:.1f} us; {} runs ({} whole tiles, {} partial, {} of them summed), K tiles per run {:.1f}'.format(
            case[0], var - 2, t_plain, len(runs), int(full.sum()), int(partial.sum()), int(last.sum()), kt.mean()))
        print('   tables {:.0f} | first loads {:.0f} | K loop {:.0f} = {:.0f} per K tile'.format(
            d(1, 0).mean(), d(2, 1).mean(), d(3, 2).mean(), (d(3, 2) / kt).mean()))
        if partial.any():
            print('   partial runs: publish {:.0f}'.format(d(4, 3)[partial].mean()))
        if last.any():
            print('   last arrivers: sum pieces {:.0f} | stage operands {:.0f} | build tile {:.0f} | stores {:.0f}'.format(
                d(5, 4)[last].mean(), d(6, 5)[last].mean(), d(7, 6)[last].mean(), d(8, 7)[last].mean()))
        if full.any():
            print('   whole tiles: stage operands {:.0f} | build tile {:.0f} | stores {:.0f}   (run total {:.0f})'.format(
                d(6, 5)[full].mean(), d(7, 6)[full].mean(), d(8, 7)[full].mean(), d(8, 0)[full].mean()))
