"""
CPU check of the per-pixel arithmetic header the HIP kernels inline (cutmix-semisup-seg_amd/csrc/pixel_math.hpp),
driven on the host by tests/hostcheck (test infrastructure) and compared with the oracle and the golden vectors.
The kernels themselves (indexing, reductions, LDS tiling) are covered by the `-m gpu` tests.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import load_golden, load_golden_json, REPO
from oracle import boxmask, losses as olosses, ema_opt

HC_DIR = os.path.join(REPO, 'tests', 'hostcheck')


@pytest.fixture(scope='module')
def hc():
    subprocess.check_call(['make', '-s', '-C', HC_DIR])
    lib = ctypes.CDLL(os.path.join(HC_DIR, '_build', 'libhostcheck.so'))
    return lib


def _p(a, ty=ctypes.c_float):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ty))


def _f32(t):
    return np.ascontiguousarray(t, dtype=np.float32)


LOSS_ID = dict(var=0, logits_var=1, logits_smoothl1=2, bce=3, kld=4)


def run_consistency(hc, l_stu, l0, l1, mask, um0, um1, H, W, align, mode, fn, tau, pp, gscale=None):
    n, c, h, w = l_stu.shape
    stats = np.zeros(3, dtype=np.float64)
    grad = np.zeros_like(l_stu) if gscale is not None else None
    hc.hc_consistency(_p(l_stu), _p(l0), _p(l1), _p(mask), _p(um0), _p(um1), n, c, h, w, H, W, int(align),
                      0 if mode == 'mix' else 1, LOSS_ID[fn], ctypes.c_float(tau), int(pp),
                      stats.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                      ctypes.c_float(0.0 if gscale is None else gscale), _p(grad))
    return stats, grad


def finalize(stats, P, tau, pp, ramp, weight):
    """mirror of cons_finalize_kernel"""
    if tau > 0:
        rate = stats[2] / P
        if pp:
            closs, gs = stats[1] / P, 1.0 / P
        else:
            closs, gs = rate * stats[0] / P, rate / P
    else:
        rate, closs, gs = float('nan'), stats[0] / P, 1.0 / P
    closs *= ramp
    return closs, rate, gs * ramp * weight, closs * weight


_LC = load_golden_json('losses_meta')


@pytest.mark.parametrize('case', _LC, ids=[c['key'] for c in _LC])
def test_consistency_pixel_math_vs_golden(hc, case):
    g = load_golden('losses')
    pre = 'C{}__'.format(case['C'])
    a = lambda n: _f32(g[pre + n])
    l_stu, l0, l1, mask, um0, um1 = a('l_stu'), a('l0_tea'), a('l1_tea'), a('mask'), a('um0'), a('um1')
    n, c, H, W = l_stu.shape
    ramp = case['ramp_val'] if case['rampup'] > 0 else 1.0
    stats, _ = run_consistency(hc, l_stu, l0, l1, mask, um0, um1, H, W, True, case['mode'], case['fn'],
                               case['conf_thresh'], case['conf_per_pixel'])
    closs, rate, gs, unsup = finalize(stats, n * H * W, case['conf_thresh'], case['conf_per_pixel'], ramp,
                                      case['cons_weight'])
    want_closs, want_unsup, want_rate = g[case['key'] + '__vals']
    assert closs == pytest.approx(want_closs, rel=1e-5, abs=1e-9)
    assert unsup == pytest.approx(want_unsup, rel=1e-5, abs=1e-9)
    if case['conf_thresh'] > 0:
        assert rate == pytest.approx(want_rate, abs=1e-7)
    _, grad = run_consistency(hc, l_stu, l0, l1, mask, um0, um1, H, W, True, case['mode'], case['fn'],
                              case['conf_thresh'], case['conf_per_pixel'], gscale=gs)
    want = g[case['key'] + '__grad']
    np.testing.assert_allclose(grad, want, rtol=1e-3, atol=2e-5 * max(1e-12, np.abs(want).max()))


@pytest.mark.parametrize('ac', [True, False])
@pytest.mark.parametrize('fn', ['var', 'kld', 'logits_smoothl1'])
def test_consistency_with_upsample_vs_oracle(hc, ac, fn):
    gen = torch.Generator().manual_seed(3)
    N, C, h, w, H, W = 2, 5, 6, 7, 41, 50
    ls = torch.randn(N, C, h, w, generator=gen) * 2
    l0 = torch.randn(N, C, h, w, generator=gen) * 3
    l1 = torch.randn(N, C, h, w, generator=gen) * 3
    um0 = (torch.rand(N, 1, H, W, generator=gen) > 0.3).float()
    um1 = (torch.rand(N, 1, H, W, generator=gen) > 0.3).float()
    m = torch.tensor(boxmask.generate_params(N, (H, W), 0.5, invert=True,
                                             rng=np.random.RandomState(2)).astype(np.float32))
    tau, pp = 0.6, True
    ls_g = ls.clone().requires_grad_(True)
    up = lambda t: olosses.upsample(t, (H, W), align_corners=ac)
    r = olosses.mix_mode_loss(up(ls_g), up(l0), up(l1), m, um0, um1, loss_fn=fn, conf_thresh=tau, conf_per_pixel=pp)
    r['unsup_loss'].backward()
    stats, _ = run_consistency(hc, _f32(ls), _f32(l0), _f32(l1), _f32(m), _f32(um0), _f32(um1), H, W, ac, 'mix', fn,
                               tau, pp)
    closs, rate, gs, unsup = finalize(stats, N * H * W, tau, pp, 1.0, 1.0)
    assert closs == pytest.approx(float(r['consistency_loss'].detach()), rel=2e-5)
    assert rate == pytest.approx(float(r['conf_rate']), abs=2e-6)
    _, grad = run_consistency(hc, _f32(ls), _f32(l0), _f32(l1), _f32(m), _f32(um0), _f32(um1), H, W, ac, 'mix', fn,
                              tau, pp, gscale=gs)
    want = ls_g.grad.numpy()
    np.testing.assert_allclose(grad, want, rtol=5e-4, atol=5e-6 * np.abs(want).max())


@pytest.mark.parametrize('C', [21, 2])
def test_ce_pixel_math_vs_golden(hc, C):
    g = load_golden('losses')
    pre = 'C{}__'.format(C)
    l = _f32(g[pre + 'l_stu'])
    y = np.ascontiguousarray(g[pre + 'labels'].astype(np.int64))
    n, c, H, W = l.shape
    stats = np.zeros(2)
    hc.hc_ce(_p(l), _p(y, ctypes.c_int64), 255, n, c, H, W, H, W, 1,
             stats.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_float(0), None)
    assert stats[0] / stats[1] == pytest.approx(float(g[pre + 'ce__val']), rel=2e-6)
    grad = np.zeros_like(l)
    hc.hc_ce(_p(l), _p(y, ctypes.c_int64), 255, n, c, H, W, H, W, 1,
             stats.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_float(1.0 / stats[1]), _p(grad))
    np.testing.assert_allclose(grad, g[pre + 'ce__grad'], rtol=2e-4, atol=1e-8)


@pytest.mark.parametrize('ac', [True, False])
def test_bilinear_taps_vs_golden(hc, ac):
    g = load_golden('losses')
    lo = _f32(g['up__lo'])
    hi = np.zeros((2, 5, 33, 41), dtype=np.float32)
    hc.hc_upsample(_p(lo), _p(hi), 10, 5, 7, 33, 41, int(ac))
    np.testing.assert_allclose(hi, g['up__hi_ac{}'.format(int(ac))], rtol=1e-5, atol=1e-6)


_BM = load_golden_json('boxmask_meta')


@pytest.mark.parametrize('case', [c for c in _BM['cases'] if c['shape'] in ([32, 32], [33, 47], [321, 321])],
                         ids=lambda c: c['key'])
def test_box_membership_bit_exact(hc, case):
    import hashlib
    kw = dict(_BM['flagsets'][case['flagset']])
    pr = kw.pop('prop_range')
    pr = tuple(pr) if isinstance(pr, list) else pr
    invert = kw.pop('invert')
    shape = tuple(case['shape'])
    with np.errstate(all='ignore'):
        rects = boxmask.draw_rects(_BM['n'], shape, pr, rng=np.random.RandomState(case['seed']), **kw)
    rng_ = np.ascontiguousarray(boxmask.rects_to_ranges(rects, shape))
    out = np.zeros((_BM['n'], 1) + shape, dtype=np.float32)
    hc.hc_box_mask(_p(rng_, ctypes.c_int32), _BM['n'], rng_.shape[1], shape[0], shape[1], int(invert), _p(out))
    assert hashlib.sha256(out.astype(np.uint8).tobytes()).hexdigest() == case['sha256']


@pytest.mark.parametrize('alpha', [0.99, 0.5, 0.999])
def test_ema_three_roundings_bit_exact(hc, alpha):
    g = load_golden('ema')
    k = 'conv.weight'
    tag = 'a{}'.format(alpha)
    t = g['{}__init__{}'.format(tag, k)].copy().reshape(-1)
    for step in range(3):
        s = np.ascontiguousarray(g['{}__src{}__{}'.format(tag, step, k)].reshape(-1))
        hc.hc_ema(_p(t), _p(s), ctypes.c_size_t(t.size), ctypes.c_float(alpha), ctypes.c_float(1.0 - alpha))
        np.testing.assert_array_equal(t, g['{}__tgt{}__{}'.format(tag, step, k)].reshape(-1))
