#!/usr/bin/env python
"""Follow-up of tools/cu_mask_probe.py: (1) the full map mask bit -> physical CU, (2) launch time of one conv8 layer on plain /
masked streams without any event, (3) cost of a cross-stream hop between plain and between masked streams."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import ctypes
from cu_mask_probe_lib import masked_stream, where, DEV, ops, hip

print('== 1: bit -> (xcc, se, sh, cu)', flush=True)
m = {}
for b in range(256):
    st = masked_stream([b])
    m[b] = where(st, 1, spin_us=1)[0]
    hip.hipStreamDestroy(ctypes.c_void_p(st.cuda_stream))      # (a masked stream owns a hardware queue: 256 of them crash the runtime)
    del st
for b0 in range(0, 256, 8):
    print('  bits {:3d}..{:3d}: {}'.format(b0, b0 + 7, ' '.join(str(m[b]) for b in range(b0, b0 + 8))))
sys.stdout.flush()
print('  distinct physical CUs reachable: {}'.format(len(set(m.values()))), flush=True)

N, H, W = 20, 41, 41
g = torch.Generator(device=DEV).manual_seed(0)
a1 = torch.randn(N, H, W, 256, generator=g, device=DEV).bfloat16()
a2 = torch.empty_like(a1)
w2 = (torch.randn(9, 256, 256, generator=g, device=DEV) * 0.02).bfloat16()
s, bz = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
T3 = ops.conv_taps(3, 3, 2, 2)
tiny = torch.zeros(64, device=DEV)


def conv2():
    ops.conv_igemm(a1, w2, T3, scale=s, bias=bz, relu=True, out=a2)


def gate_time(streams, enqueue):
    best = 1e30
    main = torch.cuda.current_stream(DEV)
    for _ in range(4):
        torch.cuda.synchronize()
        e0, e1, gate = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
        torch.cuda._sleep(40_000_000)
        e0.record(main); gate.record(main)
        for st in streams:
            st.wait_event(gate)
        enqueue()
        for st in streams:
            main.wait_stream(st)
        e1.record(main)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


print('== 2: 40 launches of the layer-3 3x3 (132 tiles) back to back on ONE stream, us per launch', flush=True)
conv2(); torch.cuda.synchronize()
cands = [('plain', torch.cuda.Stream(device=DEV)), ('mask all 256 bits', masked_stream(range(256))),
         ('mask bits 0..131', masked_stream(range(132))), ('mask bits 0..143', masked_stream(range(144))),
         ('mask bits 0..199', masked_stream(range(200)))]
for name, st in cands:
    def enq():
        with torch.cuda.stream(st):
            for _ in range(40):
                conv2()
    print('  {:<22s} {:8.1f}'.format(name, gate_time([st], enq) / 40))

print('== 3: 40 hops a -> b -> a (a tiny kernel on each side), us per hop', flush=True)
for name, mk in (('plain', lambda: torch.cuda.Stream(device=DEV)), ('masked (all bits)', lambda: masked_stream(range(256))),
                 ('masked (halves)', None)):
    if mk is None:
        sa, sb = masked_stream(range(128)), masked_stream(range(128, 256))
    else:
        sa, sb = mk(), mk()

    def enq():
        for _ in range(40):
            with torch.cuda.stream(sa):
                tiny.add_(1.0)
            sb.wait_stream(sa)
            with torch.cuda.stream(sb):
                tiny.add_(1.0)
            sa.wait_stream(sb)
    print('  {:<22s} {:8.1f}'.format(name, gate_time([sa, sb], enq) / 80))
