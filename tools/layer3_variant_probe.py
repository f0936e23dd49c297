#!/usr/bin/env python
"""Layer-3 forward of the student || teacher pair (tools/layer3_pair_probe.py) with the two K-deep convolutions of a bottleneck on
different kernels: variant 0 = the default route (eight-phase conv8), 99 = the 128 x 128 family, tile codes 256 / 1128 of that
family. us per bottleneck, one chain / two chains.
    python tools/layer3_variant_probe.py [N]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from cu_mask_probe_lib import DEV, ops
import layer3_pair_probe as L          # (runs its own table first)

A, B, pa, pb = L.A, L.B, L.pa, L.pb


def mk(kind, which):
    kw = {'8': dict(variant=0), '128': dict(variant=99), 't256': dict(variant=99, tile=256), 't1128': dict(variant=99, tile=1128)}[kind]
    if which == 1:
        return lambda n, i: ops.conv_igemm(n['x'][i & 1], n['w1'], L.T1, scale=n['s256'], bias=n['b256'], relu=True, out=n['a1'], **kw)
    return lambda n, i: ops.conv_igemm(n['a1'], n['w2'], L.T3, scale=n['s256'], bias=n['b256'], relu=True, out=n['a2'], **kw)


print('== kernels of conv1 / conv2 (expansion always on the balanced 128 x 128 launch); us per bottleneck')
print('   {:<22s} {:>10s} {:>12s}'.format('conv1 / conv2', 'one chain', 'two chains'))
for k1, k2 in (('8', '8'), ('128', '8'), ('8', '128'), ('128', '128'), ('t256', '8'), ('t256', 't256'), ('t1128', '8')):
    try:
        c1, c2 = mk(k1, 1), mk(k2, 2)
        c1(A, 0); c2(A, 0); torch.cuda.synchronize()
        parts = (c1, c2, L.conv3)
        one = L.timed(L.chain(A, pa, parts), [pa]) / L.BLOCKS
        ga, gb = L.chain(A, pa, parts), L.chain(B, pb, parts)
        two = L.timed(lambda: (ga(), gb()), [pa, pb]) / L.BLOCKS
        print('   {:<22s} {:10.1f} {:12.1f}'.format(k1 + ' / ' + k2, one, two))
    except Exception as e:            # noqa
        print('   {:<22s} failed: {}'.format(k1 + ' / ' + k2, str(e)[:120]))

# asymmetric pairs: the student's chain on one set of kernels, the teacher's on another -- two eight-phase launches of 132 tiles are 264
# whole-CU workgroups for 256 CUs; an eight-phase launch beside a 128 x 128 launch (four per CU) has no such collision
print('== asymmetric pairs (chain A kernels | chain B kernels); us per bottleneck pair')
for ka, kb in ((('8', '8'), ('128', '128')), (('8', '8'), ('128', '8')), (('8', '8'), ('8', '128')), (('8', '8'), ('t1128', 't1128'))):
    try:
        pa_ = (mk(ka[0], 1), mk(ka[1], 2), L.conv3)
        pb_ = (mk(kb[0], 1), mk(kb[1], 2), L.conv3)
        for f in pb_[:2]:
            f(B, 0)
        torch.cuda.synchronize()
        ga, gb = L.chain(A, pa, pa_), L.chain(B, pb, pb_)
        two = L.timed(lambda: (ga(), gb()), [pa, pb]) / L.BLOCKS
        print('   {:<12s} | {:<14s} {:10.1f}'.format(' / '.join(ka), ' / '.join(kb), two))
    except Exception as e:            # noqa
        print('   {} | {} failed: {}'.format(ka, kb, str(e)[:120]))
