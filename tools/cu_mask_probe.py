#!/usr/bin/env python
"""CU-masked streams (hipExtStreamCreateWithCUMask) for the forward passes: does confining the whole-CU eight-phase launches
(conv8: 146 KB of LDS, one workgroup per CU, 132 tiles per launch at cfg 2) to one set of CUs and the four-per-CU HBM-side
launches (256 -> 1024 expansions) to the rest shorten a layer-3 bottleneck of the student || teacher pair?

Why it might: in the step a conv8 launch takes 1.6-1.9 x its time alone (profiles/r06w_step_timeline.txt). A conv8 workgroup needs
an EMPTY CU; the other stream's expansion has ~1 052 32-KB workgroups queued that fit into any CU with a free quarter -- they win
every CU that frees up, so the conv8 launch starves until the expansion has nothing left to dispatch.

Part 1: which physical CUs a mask bit selects (tools/hwid_probe.hip).   Part 2: 23 bottlenecks of layer 3 at cfg 2 geometry for two
networks: (a) two plain streams (today), (b) four masked streams (per network: conv8 stream on S1, expansion stream on S2, events
between), (c) the same four streams unmasked (cost of the events alone), (d) each alone.
    python tools/cu_mask_probe.py"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cutmix_semisup_seg_amd import ops

DEV = torch.device('cuda:0')
hip = ctypes.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libhwid_probe.so'))
probe.hwid_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
N_CU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    words = (N_CU + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), words, arr)
    if rc != 0:
        raise RuntimeError('hipExtStreamCreateWithCUMask -> {}'.format(rc))
    return torch.cuda.ExternalStream(h.value, device=DEV)


def where(stream, blocks, threads=512, lds=146 * 1024, spin_us=40):
    out = torch.zeros(blocks * 2, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    rc = probe.hwid_launch(out.data_ptr(), blocks, threads, lds, int(spin_us * 100), stream.cuda_stream)   # wall_clock64: 100 MHz
    assert rc == 0, rc
    torch.cuda.synchronize()
    r = out.cpu().numpy().view(np.uint32).reshape(blocks, 2)
    hw, xcc = r[:, 0], r[:, 1] & 0xf
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
    return [(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(xcc, se, sh, cu)]


def describe(name, locs):
    cus = sorted(set(locs))
    per_xcc = {}
    for l in cus:
        per_xcc[l[0]] = per_xcc.get(l[0], 0) + 1
    print('  {:<34s} {:4d} workgroups on {:3d} distinct CUs; per XCC: {}'.format(name, len(locs), len(cus),
                                                                         ' '.join('{}:{}'.format(k, per_xcc[k]) for k in sorted(per_xcc))))
    return set(cus)


print('== part 1: mask bit -> physical CU   ({} CUs)'.format(N_CU))
plain = torch.cuda.Stream(device=DEV)
all_cus = describe('no mask, 256 whole-CU workgroups', where(plain, 256))
describe('no mask, 132 whole-CU workgroups', where(plain, 132))
s_lo = masked_stream(range(0, 132))
s_hi = masked_stream(range(132, 256))
lo = describe('bits 0..131, 132 workgroups', where(s_lo, 132))
hi = describe('bits 132..255, 124 workgroups', where(s_hi, 124))
print('  overlap of the two sets: {}   union: {}'.format(len(lo & hi), len(lo | hi)))
for b0 in (0, 1, 8, 32, 128):
    describe('bits {}..{} (8 bits)'.format(b0, b0 + 7), where(masked_stream(range(b0, b0 + 8)), 8))
# hypothesis H2 (bit = xcc * 32 + cu): 17 / 16 CUs of every XCC
h2 = [x * 32 + c for x in range(8) for c in range(17 if x < 4 else 16)]
s_h2 = masked_stream(h2)
describe('H2 set (xcc*32 + cu < 17/16)', where(s_h2, 132))

# ---------------------------------------------------------------------------------------------------------------- part 2
N, H, W = 20, 41, 41
BLOCKS = 23


def make_net(seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, device=DEV)
    net = {'x': [r(N, H, W, 1024).bfloat16() for _ in range(2)], 'a1': torch.empty(N, H, W, 256, dtype=torch.bfloat16, device=DEV),
           'a2': torch.empty(N, H, W, 256, dtype=torch.bfloat16, device=DEV),
           'w1': (r(1, 256, 1024) * 0.03).bfloat16(), 'w2': (r(9, 256, 256) * 0.02).bfloat16(), 'w3': (r(1, 1024, 256) * 0.05).bfloat16(),
           's256': torch.ones(256, device=DEV), 'b256': torch.zeros(256, device=DEV), 's1024': torch.full((1024,), 0.5, device=DEV),
           'b1024': torch.zeros(1024, device=DEV), 'bits': torch.empty(N, H, W, 128, dtype=torch.uint8, device=DEV)}
    return net


T1, T3 = ops.conv_taps(1, 1, 1, 0), ops.conv_taps(3, 3, 2, 2)


def conv1(n, i):
    ops.conv_igemm(n['x'][i & 1], n['w1'], T1, scale=n['s256'], bias=n['b256'], relu=True, out=n['a1'])


def conv2(n, i):
    ops.conv_igemm(n['a1'], n['w2'], T3, scale=n['s256'], bias=n['b256'], relu=True, out=n['a2'])


def conv3(n, i):
    ops.conv_igemm(n['a2'], n['w3'], T1, scale=n['s1024'], bias=n['b1024'], res=n['x'][i & 1], relu=True, out=n['x'][(i + 1) & 1],
                   mask_bits_out=n['bits'])


def timed(enqueue, streams, reps=3):
    """enqueue(): puts the whole pass on `streams` behind a gate; returns us of GPU time from the gate's opening to the join."""
    best = None
    main = torch.cuda.current_stream(DEV)
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        e0, e1, gate = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
        torch.cuda._sleep(60_000_000)                     # ~25-30 ms: the host enqueues everything meanwhile
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


def chain_plain(net, s):
    def go():
        with torch.cuda.stream(s):
            for i in range(BLOCKS):
                conv1(net, i); conv2(net, i); conv3(net, i)
    return go


def chain_split(net, sm, sh):
    """conv8 launches on `sm`, the expansion on `sh`, events between (two hops per bottleneck)."""
    def go():
        for i in range(BLOCKS):
            with torch.cuda.stream(sm):
                conv1(net, i); conv2(net, i)
            sh.wait_stream(sm)
            with torch.cuda.stream(sh):
                conv3(net, i)
            sm.wait_stream(sh)
    return go


def both(*gos):
    def go():
        # interleave the two networks' enqueues bottleneck by bottleneck is not possible with closures over whole chains; the
        # gate makes the host order irrelevant (everything is queued before the GPU starts)
        for g in gos:
            g()
    return go


A, B = make_net(1), make_net(2)
for f in (conv1, conv2, conv3):
    f(A, 0); f(B, 0)
torch.cuda.synchronize()
pa, pb = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
flops_block = 2.0 * N * H * W * (1024 * 256 + 9 * 256 * 256 + 256 * 1024)


def report(name, us, nets):
    print('  {:<58s} {:8.1f} us per bottleneck{}   {:5.2f} PFLOP/s'.format(name, us / BLOCKS, ' pair' if nets == 2 else '     ',
                                                                       nets * flops_block * BLOCKS / us * 1e-9))


print('== part 2: layer 3 forward at cfg 2 geometry (N = 20, 41 x 41), {} bottlenecks'.format(BLOCKS))
report('one network alone, plain stream', timed(chain_plain(A, pa), [pa]), 1)
report('two networks, two plain streams (today)', timed(both(chain_plain(A, pa), chain_plain(B, pb)), [pa, pb]), 2)
q = [torch.cuda.Stream(device=DEV) for _ in range(4)]
report('two networks, four plain streams + events', timed(both(chain_split(A, q[0], q[1]), chain_split(B, q[2], q[3])), q), 2)
for label, s1_bits, s2_bits in (('interleaved bits (0..131 | 132..255)', list(range(132)), list(range(132, 256))),
                                ('blocked bits (xcc*32+cu)', h2, [b for b in range(256) if b not in set(h2)]),
                                ('interleaved, 136 | 120', list(range(136)), list(range(136, 256))),
                                ('interleaved, 132 | all', list(range(132)), list(range(256)))):
    m = [masked_stream(s1_bits), masked_stream(s2_bits), masked_stream(s1_bits), masked_stream(s2_bits)]
    report('one network alone, masked pair: ' + label, timed(chain_split(A, m[0], m[1]), m[:2]), 1)
    report('two networks, four masked streams: ' + label, timed(both(chain_split(A, m[0], m[1]), chain_split(B, m[2], m[3])), m), 2)
    # three queues: ONE conv8 stream for both networks (A1 A2 B1 B2 ...), an expansion stream per network
    def three():
        for i in range(BLOCKS):
            for net, sh in ((A, m[1]), (B, m[3])):
                m[0].wait_stream(sh)
                with torch.cuda.stream(m[0]):
                    conv1(net, i); conv2(net, i)
                sh.wait_stream(m[0])
                with torch.cuda.stream(sh):
                    conv3(net, i)
    report('two networks, ONE masked conv8 stream + two expansion streams', timed(three, [m[0], m[1], m[3]]), 2)
