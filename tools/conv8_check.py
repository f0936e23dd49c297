#!/usr/bin/env python
"""The eight-phase 256 x 256 convolution (csrc/conv8.hip, variants 90 = whole tiles / 91 = persistent + stream-K) against
the default kernel (variant 0 with CMS_CONV8 unset): every epilogue kind, odd geometries, repeated launches (a race in the
counted-wait pipeline or in the slab hand-off shows up as a mismatch that comes and goes), two streams at once; then the
timing of the DeepLab v2 layer shapes of BASELINE configs[1] / configs[2].
    python tools/conv8_check.py [check] [time] [shape-name substrings ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import ops

DEV = 'cuda:0'
PERF = [
    # name, N, H, W, Cin, Cout, k, dil
    ('c2 l3 1x1 1024->256', 20, 41, 41, 1024, 256, 1, 1),
    ('c2 l3 3x3d2 256->256', 20, 41, 41, 256, 256, 3, 2),
    ('c2 l4 1x1 2048->512', 20, 41, 41, 2048, 512, 1, 1),
    ('c2 l4 3x3d4 512->512', 20, 41, 41, 512, 512, 3, 4),
    ('c2 l4 1x1 512->2048', 20, 41, 41, 512, 2048, 1, 1),
    ('c2 l3 1x1 256->1024', 20, 41, 41, 256, 1024, 1, 1),
    ('c3 l3 1x1 1024->256', 8, 65, 129, 1024, 256, 1, 1),
    ('c3 l3 3x3d2 256->256', 8, 65, 129, 256, 256, 3, 2),
    ('c3 l4 1x1 2048->512', 8, 65, 129, 2048, 512, 1, 1),
    ('c3 l4 3x3d4 512->512', 8, 65, 129, 512, 512, 3, 4),
    ('c3 l4 1x1 512->2048', 8, 65, 129, 512, 2048, 1, 1),
    ('c3 l3 1x1 256->1024', 8, 65, 129, 256, 1024, 1, 1),
]
SMALL = [
    ('one partial tile 3x3', 1, 9, 13, 128, 256, 3, 1),
    ('odd K tiles (9)', 2, 17, 23, 64, 256, 3, 2),
    ('odd K tiles (5 taps x 1)', 3, 19, 21, 64, 512, 1, 1),
    ('tiles 5 x 3', 3, 20, 21, 192, 768, 3, 3),
]
EPILOGUES = ['relu', 'res_relu', 'plain', 'dgrad_mask', 'dgrad_both', 'dgrad_add']


def make(case, seed=0):
    name, N, H, W, Cin, Cout, k, dil = case
    g = torch.Generator(device=DEV).manual_seed(seed)
    pad = dil * (k - 1) // 2
    x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
    wp = (torch.randn(k * k, Cout, Cin, generator=g, device=DEV) * (2.0 / (Cin * k * k)) ** 0.5).bfloat16()
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    bias = torch.randn(Cout, generator=g, device=DEV) * 0.1
    res = torch.randn(N, H, W, Cout, generator=g, device=DEV).bfloat16()
    msk = torch.randn(N, H, W, Cout, generator=g, device=DEV).bfloat16()
    return x, wp, ops.conv_taps(k, k, dil, pad), scale, bias, res, msk


def run(t, epi, variant, out=None):
    x, wp, taps, scale, bias, res, msk = t
    if epi == 'relu':
        return ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, relu=True, out=out, variant=variant)
    if epi == 'res_relu':
        return ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, res=res, relu=True, out=out, variant=variant)
    if epi == 'plain':
        return ops.conv_igemm(x, wp, taps, out=out, variant=variant)
    if epi == 'dgrad_mask':
        return ops.conv_igemm(x, wp, taps, mode=1, mask_src=msk, out=out, variant=variant)
    if epi == 'dgrad_both':
        return ops.conv_igemm(x, wp, taps, mode=1, mask_src=msk, res=res, out=out, variant=variant)
    return ops.conv_igemm(x, wp, taps, mode=1, res=res, out=out, variant=variant)


def within_one_ulp(o, ref):
    """bf16 outputs of two fp32 summation orders: equal, or neighbours in bf16 (one rounding flipped)."""
    a, b = o.float(), ref.float()
    ulp = torch.maximum(a.abs(), b.abs()) * 2.0 ** -7          # >= one bf16 ulp of the larger magnitude
    # + the fp32 summation noise itself where the result cancels to ~0 (a ReLU input of 1e-7 vs -1e-7)
    return bool(((a - b).abs() <= ulp + 2e-6 * float(b.abs().max())).all())


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
    return e0.elapsed_time(e1) / iters * 1e3       # us


def check(cases, epis, reps=3):
    bad = []
    for case in cases:
        t = make(case)
        for epi in epis:
            ref = run(t, epi, 99)
            torch.cuda.synchronize()
            line = '{:<26s} {:<11s}'.format(case[0], epi)
            for var in (90, 91):
                outs = []
                for _ in range(reps):
                    out = torch.full_like(ref, 7.0)
                    run(t, epi, var, out=out)
                    outs.append(out)
                torch.cuda.synchronize()
                exact = all(torch.equal(o, ref) for o in outs)
                same = all(torch.equal(o, outs[0]) for o in outs[1:])
                d = max(float((o.float() - ref.float()).abs().max()) for o in outs)
                frac = max(float((o != ref).float().mean()) for o in outs)
                close = all(within_one_ulp(o, ref) for o in outs)
                line += '  v{}: {} maxdiff {:.3g} differing {:.2e}{}'.format(
                    var, 'EXACT' if exact else ('close' if close else 'WRONG'), d, frac, '' if same else ' NOT-REPRODUCIBLE')
                if not close or not same or (var == 90 and not exact):
                    bad.append((case[0], epi, var, d, frac, same))
            print(line, flush=True)
    return bad


def check_two_streams(case, epi='res_relu', rounds=6):
    """Two streams launch the stream-K variant at the same time (each stream has its own workspace): what the student ||
    teacher passes of the step do."""
    t0, t1 = make(case, 1), make(case, 2)
    ref0, ref1 = run(t0, epi, 99), run(t1, epi, 99)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    bad = 0
    for _ in range(rounds):
        o0, o1 = torch.full_like(ref0, 7.0), torch.full_like(ref1, 7.0)
        torch.cuda.synchronize()
        for _ in range(4):
            with torch.cuda.stream(s0):
                run(t0, epi, 91, out=o0)
            with torch.cuda.stream(s1):
                run(t1, epi, 91, out=o1)
        torch.cuda.synchronize()
        ok = within_one_ulp(o0, ref0) and within_one_ulp(o1, ref1)
        bad += 0 if ok else 1
    print('two streams, {} x4 launches of {:<26s}: {}'.format(rounds, case[0], 'ok' if bad == 0 else '{} BAD rounds'.format(bad)),
          flush=True)
    return bad


def bench(cases):
    print('{:<26s}{:>10s}{:>10s}{:>10s}   PF/s: default / 90 / 91'.format('shape', 'default', 'v90', 'v91'))
    for case in cases:
        name, N, H, W, Cin, Cout, k, dil = case
        t = make(case)
        flops = 2.0 * N * H * W * Cout * Cin * k * k
        for epi in ('relu',) if Cout < 1024 else ('res_relu',):
            out = run(t, epi, 99)
            ts = [timeit(lambda v=v: run(t, epi, v, out=out)) for v in (99, 90, 91)]
            print('{:<26s}{:>10.1f}{:>10.1f}{:>10.1f}   {:.2f} / {:.2f} / {:.2f}'.format(
                name + (' +res' if epi == 'res_relu' else ''), ts[0], ts[1], ts[2], *[flops / x / 1e9 for x in ts]), flush=True)


if __name__ == '__main__':
    args = sys.argv[1:]
    do_check = 'check' in args or not any(a in ('check', 'time') for a in args)
    do_time = 'time' in args or not any(a in ('check', 'time') for a in args)
    subs = [a for a in args if a not in ('check', 'time')]
    perf = [c for c in PERF if not subs or any(s in c[0] for s in subs)]
    bad = []
    if do_check:
        bad += check(SMALL, EPILOGUES)
        bad += check(perf[:4] + perf[6:8], ['relu', 'dgrad_both'])
        bad += check(perf[4:6], ['res_relu'])
        nb = check_two_streams(PERF[1]) + check_two_streams(PERF[7])
        print('MISMATCHES:', bad if bad else 'none', '| two-stream bad rounds:', nb, flush=True)
    if do_time:
        bench(perf)
