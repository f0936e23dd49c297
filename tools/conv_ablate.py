import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import ops
DEV = 'cuda:0'
N = 20
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print('{:<22s} {:>9s} {:>9s} {:>9s} {:>9s} {:>9s} {:>9s}'.format('shape', 'full', 'no_mfma', 'no_loads', 'regstage', '8w_128', '8w_256'))
for name, H, W, Cin, Cout, k, dil in [('l3 1x1 256->1024', 41, 41, 256, 1024, 1, 1), ('l3 1x1 1024->256', 41, 41, 1024, 256, 1, 1),
                                      ('l3 3x3d2 256->256', 41, 41, 256, 256, 3, 2), ('l4 3x3d4 512->512', 41, 41, 512, 512, 3, 4),
                                      ('l4 1x1 1024->2048', 41, 41, 1024, 2048, 1, 1), ('l1 1x1 64->256', 81, 81, 64, 256, 1, 1)]:
    g = torch.Generator(device=DEV).manual_seed(0)
    pad = dil * (k - 1) // 2
    x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
    wp = (torch.randn(k * k, Cout, Cin, generator=g, device=DEV) * 0.05).bfloat16()
    scale, bias = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    out = torch.empty(N, H, W, Cout, dtype=torch.bfloat16, device=DEV)
    taps = ops.conv_taps(k, k, dil, pad)
    ts = [timeit(lambda v=v: ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, relu=True, out=out, variant=v)) for v in (0, 2, 3, 1)]
    ts.append(timeit(lambda: ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, relu=True, out=out, tile=1128)))
    ts.append(timeit(lambda: ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, relu=True, out=out, tile=256)))
    ref = ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, relu=True).float()
    for tl in (1128, 256):
        got = ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, relu=True, tile=tl).float()
        assert float((got - ref).abs().max()) <= 1e-2 * float(ref.abs().max()), (name, tl)
    print('{:<22s} {:9.1f} {:9.1f} {:9.1f} {:9.1f} {:9.1f} {:9.1f}'.format(name, *ts))
