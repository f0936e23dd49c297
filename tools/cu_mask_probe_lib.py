#!/usr/bin/env python
"""(helpers shared by tools/cu_mask_probe*.py)
CU-masked streams (hipExtStreamCreateWithCUMask) for the forward passes: does confining the whole-CU eight-phase launches
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


