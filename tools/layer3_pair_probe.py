#!/usr/bin/env python
"""Where do the forward passes of layer 3 lose their overlap? 23 bottlenecks at cfg 2 geometry (N = 20, 41 x 41) for two networks
on two plain streams, with parts of the bottleneck removed: all three convolutions (the step), only the two eight-phase launches
(conv1 1x1 1024->256, conv2 3x3 256->256), only the expansion (1x1 256->1024 + residual + ReLU + mask bits), and each alone.
If the pair of conv8-only chains takes ~ the time of ONE chain, the 132-tile launches share the machine well and the loss is in
how the expansion mixes in; if it takes 2 x, the whole-CU launches serialise (264 tiles for 256 CUs).
    python tools/layer3_pair_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from cu_mask_probe_lib import DEV, ops

N, H, W, BLOCKS = 20, 41, 41, 23
if len(sys.argv) > 1:
    N = int(sys.argv[1])


def make_net(seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, device=DEV)
    return {'x': [r(N, H, W, 1024).bfloat16() for _ in range(2)], 'a1': torch.empty(N, H, W, 256, dtype=torch.bfloat16, device=DEV),
            'a2': torch.empty(N, H, W, 256, dtype=torch.bfloat16, device=DEV),
            'w1': (r(1, 256, 1024) * 0.03).bfloat16(), 'w2': (r(9, 256, 256) * 0.02).bfloat16(), 'w3': (r(1, 1024, 256) * 0.05).bfloat16(),
            's256': torch.ones(256, device=DEV), 'b256': torch.zeros(256, device=DEV), 's1024': torch.full((1024,), 0.5, device=DEV),
            'b1024': torch.zeros(1024, device=DEV), 'bits': torch.empty(N, H, W, 128, dtype=torch.uint8, device=DEV)}


T1, T3 = ops.conv_taps(1, 1, 1, 0), ops.conv_taps(3, 3, 2, 2)


def conv1(n, i):
    ops.conv_igemm(n['x'][i & 1], n['w1'], T1, scale=n['s256'], bias=n['b256'], relu=True, out=n['a1'])


def conv2(n, i):
    ops.conv_igemm(n['a1'], n['w2'], T3, scale=n['s256'], bias=n['b256'], relu=True, out=n['a2'])


def conv3(n, i):
    ops.conv_igemm(n['a2'], n['w3'], T1, scale=n['s1024'], bias=n['b1024'], res=n['x'][i & 1], relu=True, out=n['x'][(i + 1) & 1],
                   mask_bits_out=n['bits'])


def timed(enqueue, streams, reps=3):
    best = None
    main = torch.cuda.current_stream(DEV)
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        e0, e1, gate = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
        torch.cuda._sleep(60_000_000)
        e0.record(main)
        gate.record(main)
        for s in streams:
            s.wait_event(gate)
        enqueue()
        for s in streams:
            main.wait_stream(s)
        e1.record(main)
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3
        best = t if best is None else min(best, t)
    return best


def chain(net, s, parts):
    def go():
        with torch.cuda.stream(s):
            for i in range(BLOCKS):
                for f in parts:
                    f(net, i)
    return go


A, B = make_net(1), make_net(2)
for f in (conv1, conv2, conv3):
    f(A, 0); f(B, 0)
torch.cuda.synchronize()
pa, pb = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
tiles = (N * H * W + 255) // 256
print('== layer 3 forward, N = {} ({} tiles of 256 pixels per eight-phase launch), {} bottlenecks; us per bottleneck'.format(N, tiles, BLOCKS))
print('   {:<34s} {:>10s} {:>12s} {:>10s}'.format('launches of a bottleneck', 'one chain', 'two chains', 'two / one'))
for name, parts in (('conv1 + conv2 + expansion (step)', (conv1, conv2, conv3)), ('conv1 + conv2 (eight-phase only)', (conv1, conv2)),
                    ('conv1 only', (conv1,)), ('conv2 only', (conv2,)), ('expansion only', (conv3,)),
                    ('conv2 + expansion', (conv2, conv3)), ('conv1 + expansion', (conv1, conv3))):
    one = timed(chain(A, pa, parts), [pa]) / BLOCKS
    ga, gb = chain(A, pa, parts), chain(B, pb, parts)
    two = timed(lambda: (ga(), gb()), [pa, pb]) / BLOCKS
    print('   {:<34s} {:10.1f} {:12.1f} {:10.2f}'.format(name, one, two, two / one))
# the out-of-phase arrangement the step settles into: one chain of eight-phase launches beside one chain of expansions
ga, gb = chain(A, pa, (conv1, conv2)), chain(B, pb, (conv3, conv3))
print('   {:<34s} {:>10s} {:12.1f}'.format('[conv1 + conv2] || [2 x expansion]', '', timed(lambda: (ga(), gb()), [pa, pb]) / BLOCKS))
ga, gb = chain(A, pa, (conv2,)), chain(B, pb, (conv3,))
print('   {:<34s} {:>10s} {:12.1f}'.format('[conv2] || [expansion]', '', timed(lambda: (ga(), gb()), [pa, pb]) / BLOCKS))
ga, gb = chain(A, pa, (conv1,)), chain(B, pb, (conv3,))
print('   {:<34s} {:>10s} {:12.1f}'.format('[conv1] || [expansion]', '', timed(lambda: (ga(), gb()), [pa, pb]) / BLOCKS))
