"""Is the python-launched timing harness of tools/conv_*.py host-bound for the short layers? Times (a) a trivially small
convolution in that harness (= the host floor per launch), (b) the layer launches through the harness, (c) the same launches
recorded in a program and replayed from C++ (launch cost ~2 us): GPU time per launch without the python wrapper."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import ops
DEV = 'cuda:0'


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def mk(N, H, W, Cin, Cout, k, dil):
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
    wp = (torch.randn(k * k, Cout, Cin, generator=g, device=DEV) * 0.05).bfloat16()
    out = torch.empty(N, H, W, Cout, dtype=torch.bfloat16, device=DEV)
    return x, wp, out, ops.conv_taps(k, k, dil, dil * (k - 1) // 2)


x, wp, out, taps = mk(1, 7, 9, 64, 128, 1, 1)
print('host floor of the python harness (tiny convolution): %.1f us per launch' % timeit(lambda: ops.conv_igemm(x, wp, taps, out=out)))
print('%-22s %10s %10s %10s %10s %10s %10s' % ('shape', 'py full', 'py nomfma', 'py noload', 'prog full', 'prog nomfma', 'prog noload'))
for name, H, W, Cin, Cout, k, dil in [('l3 1x1 256->1024', 41, 41, 256, 1024, 1, 1), ('l3 1x1 1024->256', 41, 41, 1024, 256, 1, 1),
                                      ('l3 3x3d2 256->256', 41, 41, 256, 256, 3, 2), ('l4 3x3d4 512->512', 41, 41, 512, 512, 3, 4),
                                      ('l1 1x1 64->256', 81, 81, 64, 256, 1, 1)]:
    x, wp, out, taps = mk(20, H, W, Cin, Cout, k, dil)
    py = [timeit(lambda v=v: ops.conv_igemm(x, wp, taps, out=out, variant=v)) for v in (0, 2, 3)]
    pr = []
    for v in (0, 2, 3):
        prog = ops.Program()
        with ops.recording(prog, [torch.cuda.current_stream()]):
            for _ in range(50):
                ops.conv_igemm(x, wp, taps, out=out, variant=v)
        pr.append(timeit(lambda: prog.run([torch.cuda.current_stream()]), iters=4) / 50)
    print('%-22s %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f' % ((name,) + tuple(py) + tuple(pr)))
