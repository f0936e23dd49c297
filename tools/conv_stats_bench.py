"""Cost of the BatchNorm statistics in the convolution epilogues, launches alone on the GPU (round 5):
forward launches with / without cms_conv_desc.stats_out, data gradients with / without bstats_*; us per launch (median of 5 x 50)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__  # noqa: F401  (package alias)
from cutmix_semisup_seg_amd import ops

DEV = 'cuda:0'


def timeit(fn, reps=50, rounds=5):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1000.0 / reps)
    return float(np.median(out))


def main():
    g = torch.Generator().manual_seed(0)
    N, H, W = 20, 41, 41
    cases = [('l3 conv1 1x1 1024->256 (conv8)', 1024, 256, 1, 1), ('l3 conv2 3x3 256->256 d2 (conv8)', 256, 256, 3, 2),
             ('l3 conv3 1x1 256->1024 (mixed/128)', 256, 1024, 1, 1), ('l4 conv1 1x1 2048->512 (conv8)', 2048, 512, 1, 1),
             ('l4 conv3 1x1 512->2048 (128)', 512, 2048, 1, 1)]
    for name, cin, cout, k, dil in cases:
        x = (torch.randn(N, H, W, cin, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
        w = (torch.randn(k * k, cout, cin, generator=g) * (1.0 / np.sqrt(cin * k * k))).to(torch.bfloat16).to(DEV)
        taps = ops.conv_taps(k, k, dil, dil * (k // 2))
        out = torch.empty(N, H, W, cout, dtype=torch.bfloat16, device=DEV)
        for G in (1, 2):
            st = {'groups': G}
            ops.conv_igemm(x, w, taps, out=out, stats=st)

            def with_stats():
                ops.conv_igemm(x, w, taps, out=out, stats={'groups': G})
            t0 = timeit(lambda: ops.conv_igemm(x, w, taps, out=out))
            t1 = timeit(with_stats)
            print('fwd  %-36s G=%d tile rows %3d: plain %7.1f us, + statistics %7.1f us (%+.1f)' % (name, G, st['tile_rows'], t0, t1, t1 - t0))
        # data gradient writing the gradient of THIS layer's input unit: K = cout channels -> cin channels
        du = (torch.randn(N, H, W, cout, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
        wT = (torch.randn(k * k, cin, cout, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
        u = (torch.randn(N, H, W, cin, generator=g)).to(torch.bfloat16).to(DEV)
        bits = torch.randint(0, 256, (N * H * W * cin // 8,), generator=g, dtype=torch.uint8).to(DEV)
        mean, rstd = torch.zeros(2 * cin, device=DEV), torch.ones(2 * cin, device=DEV)
        dout = torch.empty(N, H, W, cin, dtype=torch.bfloat16, device=DEV)
        ntaps = [(-dy, -dx) for dy, dx in taps]
        st = {'groups': 2, 'u': u, 'mean': mean, 'rstd': rstd, 'bits': bits}
        ops.conv_igemm(du, wT, ntaps, mode=1, out=dout, stats=st)
        t0 = timeit(lambda: ops.conv_igemm(du, wT, ntaps, mode=1, out=dout))
        t1 = timeit(lambda: ops.conv_igemm(du, wT, ntaps, mode=1, out=dout, stats={'groups': 2, 'u': u, 'mean': mean, 'rstd': rstd, 'bits': bits}))
        print('bwd  %-36s G=2 tile rows %3d: plain %7.1f us, + statistics %7.1f us (%+.1f)' % (name, st['tile_rows'], t0, t1, t1 - t0))


if __name__ == '__main__':
    main()
