#!/usr/bin/env python
"""What would a GROUPED weight-gradient launch be worth? The weight gradients of K bottlenecks issued at once: per-layer
launches of cms_conv_wgrad spread over S streams with a small split-K factor, so that the machine sees the workgroups of many
layers together (what one grouped launch over a device-resident work list would give). Layer-3 bottlenecks of DeepLab v2 at
BASELINE configs[1] (fused batch 20, 41 x 41): 1x1 1024->256, 3x3 d2 256->256, 1x1 256->1024.
    python tools/wgrad_concurrency.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import ops

DEV = 'cuda:0'
N, H, W = 20, 41, 41
LAYERS = [(1024, 256, 1, 1), (256, 256, 3, 2), (256, 1024, 1, 1)]
NB = 8                      # bottlenecks in flight


def mk(nb):
    g = torch.Generator(device=DEV).manual_seed(0)
    jobs = []
    for b in range(nb):
        for Cin, Cout, k, dil in LAYERS:
            x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
            du = torch.randn(N, H, W, Cout, generator=g, device=DEV).bfloat16()
            dw = torch.zeros(k * k, Cout, Cin, device=DEV)
            jobs.append((du, x, ops.conv_taps(k, k, dil, dil * (k - 1) // 2), dw))
    return jobs


jobs = mk(NB)
flops = sum(2.0 * N * H * W * j[3].numel() for j in jobs)
print('%d launches, %.2f TF' % (len(jobs), flops / 1e12))
for n_streams, ks in [(1, 0), (2, 0), (4, 0), (1, 4), (4, 4), (8, 4), (8, 2), (8, 1), (12, 2), (24, 1), (24, 2)]:
    streams = [torch.cuda.Stream() for _ in range(n_streams)]

    def run():
        for i, (du, x, taps, dw) in enumerate(jobs):
            with torch.cuda.stream(streams[i % n_streams]):
                ops.conv_wgrad(du, x, taps, dw, ksplit=ks)

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in streams:
        s.wait_event(e0)
    for _ in range(5):
        run()
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    print('streams %2d  ksplit %-4s  %.3f ms per %d bottlenecks = %.3f ms per bottleneck   %.2f PF/s' % (
        n_streams, ks if ks else 'auto', t, NB, t / NB, flops / t / 1e12))

# the real thing: ONE grouped launch per kind (cms_conv_wgrad_group_*) over the same launches
for target in (0, 1024, 1536, 3072, 4096):
    gjobs = [(du, x, taps, dw, 1, None) for du, x, taps, dw in jobs]
    for _ in range(2):
        ops.conv_wgrad_group(gjobs, target_workgroups=target)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv_wgrad_group(gjobs, target_workgroups=target)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    print('GROUPED target %-5s  %.3f ms per %d bottlenecks = %.3f ms per bottleneck   %.2f PF/s' % (
        target if target else 'auto', t, NB, t / NB, flops / t / 1e12))
