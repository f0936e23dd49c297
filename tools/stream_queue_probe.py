#!/usr/bin/env python
"""Which streams share a hardware queue? Pairwise clash matrix of {default stream, 6 normal-priority streams, 4 high-priority streams}:
a pair that shares a queue runs two 0.1 ms spin kernels one after the other (round 5; ops._probe_side_streams uses the same test).
Question behind it: does a HIGH-priority stream (what TORCH_NCCL_HIGH_PRIORITY=1 gives RCCL's internal stream) ever share a queue with
the normal-priority streams of the step?"""
import sys
import torch

dev = torch.device('cuda:0')
cur = torch.cuda.current_stream(dev)
normal = [torch.cuda.Stream(device=dev) for _ in range(6)]
high = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(4)]
streams = [cur] + normal + high
names = ['default'] + ['n%d' % i for i in range(6)] + ['H%d' % i for i in range(4)]
spin = 200000


def pair_ms(a, b):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(a)
    with torch.cuda.stream(a):
        torch.cuda._sleep(spin)
    if b is not None:
        b.wait_event(e0)
        with torch.cuda.stream(b):
            torch.cuda._sleep(spin)
        a.wait_stream(b)
    e1.record(a)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for s in streams:
    with torch.cuda.stream(s):
        torch.cuda._sleep(1000)
pair_ms(cur, normal[0])
alone = pair_ms(cur, None)
print('alone %.4f ms' % alone)
n = len(streams)
print('        ' + ' '.join('%7s' % x for x in names))
for i in range(n):
    row = []
    for j in range(n):
        row.append('   .   ' if i == j else ('%7s' % ('CLASH' if pair_ms(streams[i], streams[j]) >= 1.5 * alone else '-')))
    print('%7s ' % names[i] + ' '.join(row))
