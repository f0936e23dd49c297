"""
Where the error of the bf16 (throughput) configuration is born -- CPU experiment on the oracle alone.

Runs the ResNet-101 DeepLab v2 of tests/test_gpu_hip_engine_parity.py (same weights, same inputs, cfg 2 geometry) through
oracle/deeplab2_chain.py three ways: fp32 storage, bf16 storage, and bf16 storage with the LAST k tensors kept in fp32
(`round_until`). Prints the relative error of every bottleneck's output against the fp32 chain, of the logits, and of the
`var` consistency loss / confidence rate of one CutMix iteration -- the table of DESIGN.md section 2.1.

    python tools/bf16_error_curve.py [N] [H] [W]            (test infrastructure: imports oracle/)
"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

from oracle import deeplab2_chain as ch, losses as L, boxmask as obox     # noqa: E402
import test_gpu_hip_engine_parity as T                                     # noqa: E402  (its _state: the test's weights)


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 321
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 321
    C, layers = 21, T.LAYERS
    st = T._state(layers, C)
    g = torch.Generator().manual_seed(77)
    rnd = lambda: torch.randn(N, 3, H, W, generator=g).bfloat16().float()
    x, ux0, ux1 = rnd(), rnd(), rnd()
    import mask_gen
    ranges = mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(N, (H, W), rng=np.random.RandomState(9))
    m = torch.tensor(obox.rasterise(ranges, (H, W), True).astype(np.float32))
    ones = torch.ones(N, 1, H, W)
    xm = L.paste(ux0, ux1, m)

    def run(storage, round_until=None):
        c = ch.Chain(st, C, layers, storage, round_until)
        l0 = c.forward(ux0, save=False)[0]
        l1 = c.forward(ux1, save=False)[0]
        ls, saved = c.forward(xm)
        return c, (l0, l1, ls), saved

    t0 = time.time()
    cf, lf, sf = run('fp32')
    tau = float(torch.softmax(L.upsample(lf[0], (H, W)), dim=1).max(dim=1)[0].median())

    def losses(lg):
        r = L.mix_mode_loss(L.upsample(lg[2], (H, W)), L.upsample(lg[0], (H, W)), L.upsample(lg[1], (H, W)), m, ones, ones,
                            loss_fn='var', conf_thresh=tau, conf_per_pixel=False, ramp_val=1.0, rampup=-1, cons_weight=1.0)
        return float(r['consistency_loss']), float(r['conf_rate'])

    ref_loss, ref_rate = losses(lf)
    print('fp32 chain: consistency loss %.6e  confidence rate %.4f  (tau %.4f; %.1f s)' % (ref_loss, ref_rate, tau, time.time() - t0))
    cb, lb, sb = run('bf16')
    of, ob = cf.block_outputs(sf), cb.block_outputs(sb)
    print('relative error of the INPUT of bottleneck i (bf16 storage vs fp32 storage), student pass on the pasted batch:')
    names = [u[0] for u in cb.units] + ['layer4 out']
    for i, (a, b) in enumerate(zip(ob, of)):
        print('  %2d %-12s %.3e' % (i, names[i], rel(a, b)))
    print('logits: %.3e' % rel(lb[2], lf[2]))
    bl, br = losses(lb)
    print('bf16 storage everywhere: consistency loss %.6e (rel %.3e)  rate %.4f (abs %.3e)' %
          (bl, abs(bl - ref_loss) / ref_loss, br, abs(br - ref_rate)))
    nb = len(cb.units)
    for keep in (1, 3, 6, 13, 26):
        _, lk, _ = run('bf16', nb - keep)
        kl, kr = losses(lk)
        print('fp32 storage for the last %2d bottlenecks: consistency loss rel %.3e  rate abs %.3e  logits %.3e' %
              (keep, abs(kl - ref_loss) / ref_loss, abs(kr - ref_rate), rel(lk[2], lf[2])))


if __name__ == '__main__':
    main()
