"""
GPU parity tests proper (`-m gpu`): every call goes through the C ABI of libcutmixseg_hip.so (via the ctypes wrappers
of cutmix-semisup-seg_amd/ops.py) and is compared with the CPU oracle on the same seeded inputs and with the
committed golden vectors; at BASELINE.json's full sizes, through size-independent properties.

Tolerances: bit-exact for box masks / paste / EMA / confusion matrices; fp32 losses and gradients within 1e-5..1e-4
relative (different summation order and exp/log implementations than ATen-CPU); bf16 network outputs against the
fp32 oracle on identical bf16-rounded weights within 2e-2 of the output scale.
"""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden, load_golden_json

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'GPU tests need a device'
    from cutmix_semisup_seg_amd import ops as _ops
    return _ops


def cu(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a)) if not torch.is_tensor(a) else a
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


# =========================================================================================== box masks + paste
_BM = load_golden_json('boxmask_meta')


@pytest.mark.parametrize('case', _BM['cases'], ids=[c['key'] for c in _BM['cases']])
def test_boxmask_rasterize_bit_exact(ops, case):
    import mask_gen
    kw = dict(_BM['flagsets'][case['flagset']])
    pr = kw.pop('prop_range')
    gen = mask_gen.BoxMaskGenerator(tuple(pr) if isinstance(pr, list) else pr, **kw)
    shape = tuple(case['shape'])
    ranges = gen.generate_ranges(_BM['n'], shape, rng=np.random.RandomState(case['seed']))
    t_params = ops.ranges_to_device(ranges, DEV)
    m = gen.torch_masks_from_params(t_params, shape, DEV)          # device rasterisation
    assert m.shape == (_BM['n'], 1) + shape and m.dtype == torch.float32
    m8 = m.cpu().numpy().astype(np.uint8)
    assert hashlib.sha256(m8.tobytes()).hexdigest() == case['sha256']


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 3, 33, 47), (3, 21, 41, 41), (10, 3, 321, 321), (4, 3, 512, 1024)])
def test_cutmix_paste_exact(ops, dtype, shape):
    from oracle import boxmask, losses as olosses
    n, c, h, w = shape
    g = torch.Generator().manual_seed(h * w)
    x0 = torch.randn(shape, generator=g).to(dtype)
    x1 = torch.randn(shape, generator=g).to(dtype)
    rects = boxmask.draw_rects(n, (h, w), (0.2, 0.7), n_boxes=2, rng=np.random.RandomState(7))
    ranges = boxmask.rects_to_ranges(rects, (h, w))
    m = torch.tensor(boxmask.rasterise(ranges, (h, w), True).astype(np.float32))
    want = olosses.paste(x0.float(), x1.float(), m).to(dtype)
    got = ops.cutmix_paste(cu(x0), cu(x1), ranges=ops.ranges_to_device(ranges, DEV), invert=True)
    assert torch.equal(got.cpu(), want)
    got_m = ops.cutmix_paste(cu(x0), cu(x1), mask=cu(m))
    assert torch.equal(got_m.cpu(), want)
    cut = ops.cutmix_paste(None, cu(x1), ranges=ops.ranges_to_device(ranges, DEV), invert=True)
    assert torch.equal(cut.cpu(), (x1.float() * m).to(dtype))
    # properties: pasting a tensor onto itself is the identity; complementary masks swap roles
    assert torch.equal(ops.cutmix_paste(cu(x0), cu(x0), ranges=ops.ranges_to_device(ranges, DEV)).cpu(), x0)
    inv = ops.cutmix_paste(cu(x1), cu(x0), ranges=ops.ranges_to_device(ranges, DEV), invert=False)
    assert torch.equal(inv.cpu(), want)


# =========================================================================================== consistency loss
_LC = load_golden_json('losses_meta')


def _cfg(ops, case_or_kw, align=True):
    return ops.ConsistencyConfig(mode=case_or_kw['mode'], loss_fn=case_or_kw['fn'],
                                 conf_thresh=case_or_kw['conf_thresh'], conf_per_pixel=case_or_kw['conf_per_pixel'],
                                 align_corners=align, invert=True)


@pytest.mark.parametrize('case', _LC, ids=[c['key'] for c in _LC])
def test_consistency_vs_golden(ops, case):
    g = load_golden('losses')
    pre = 'C{}__'.format(case['C'])
    a = lambda n: cu(g[pre + n], torch.float32)
    l_stu, l0, l1, mask, um0, um1 = a('l_stu'), a('l0_tea'), a('l1_tea'), a('mask'), a('um0'), a('um1')
    H, W = l_stu.shape[2:]
    ramp = case['ramp_val'] if case['rampup'] > 0 else 1.0
    cfg = _cfg(ops, case)
    l_stu.requires_grad_(True)
    unsup, closs, rate = ops.consistency_loss(l_stu, l0, l1 if case['mode'] == 'mix' else None, (H, W), cfg,
                                              mask=mask, um0=um0, um1=um1, ramp_val=ramp,
                                              cons_weight=case['cons_weight'])
    unsup.backward()
    want_closs, want_unsup, want_rate = g[case['key'] + '__vals']
    assert float(closs) == pytest.approx(want_closs, rel=2e-5, abs=1e-9)
    assert float(unsup) == pytest.approx(want_unsup, rel=2e-5, abs=1e-9)
    if case['conf_thresh'] > 0:
        assert float(rate) == pytest.approx(want_rate, abs=1e-7)
    want = g[case['key'] + '__grad']
    np.testing.assert_allclose(l_stu.grad.cpu().numpy(), want, rtol=1e-3, atol=2e-5 * max(1e-12, np.abs(want).max()))


@pytest.mark.parametrize('geo', [
    dict(N=2, C=5, h=6, w=7, H=41, W=50, ac=True),
    dict(N=2, C=21, h=41, w=41, H=321, W=321, ac=True),       # cfg 2 geometry (Pascal crop)
    dict(N=1, C=19, h=65, w=129, H=512, W=1024, ac=True),     # cfg 3 geometry (Cityscapes)
    dict(N=2, C=7, h=9, w=9, H=33, W=33, ac=False),           # generic class count, align_corners=False
    dict(N=1, C=21, h=17, w=17, H=65, W=65, ac=False),
])
@pytest.mark.parametrize('fn,mode,tau,pp', [('var', 'mix', 0.5, False), ('var', 'mix', 0.6, True),
                                            ('kld', 'cut', 0.5, True), ('bce', 'mix', 0.0, False),
                                            ('logits_var', 'cut', 0.0, False), ('logits_smoothl1', 'mix', 0.4, True)])
def test_consistency_fused_upsample_vs_oracle(ops, geo, fn, mode, tau, pp):
    from oracle import boxmask, losses as olosses
    N, C, h, w, H, W, ac = (geo[k] for k in ('N', 'C', 'h', 'w', 'H', 'W', 'ac'))
    gen = torch.Generator().manual_seed(C * H + w)
    ls = torch.randn(N, C, h, w, generator=gen) * 2
    l0 = torch.randn(N, C, h, w, generator=gen) * 3
    l1 = torch.randn(N, C, h, w, generator=gen) * 3
    um0 = (torch.rand(N, 1, H, W, generator=gen) > 0.2).float()
    um1 = (torch.rand(N, 1, H, W, generator=gen) > 0.2).float()
    rects = boxmask.draw_rects(N, (H, W), 0.5, rng=np.random.RandomState(3))
    ranges = boxmask.rects_to_ranges(rects, (H, W))
    m = torch.tensor(boxmask.rasterise(ranges, (H, W), True).astype(np.float32))
    ls_o = ls.clone().requires_grad_(True)
    up = lambda t: olosses.upsample(t, (H, W), align_corners=ac)
    kw = dict(loss_fn=fn, conf_thresh=tau, conf_per_pixel=pp, cons_weight=0.7)
    if mode == 'mix':
        r = olosses.mix_mode_loss(up(ls_o), up(l0), up(l1), m, um0, um1, **kw)
    else:
        r = olosses.cut_mode_loss(up(ls_o), up(l0), m, um0, **kw)
    r['unsup_loss'].backward()
    cfg = ops.ConsistencyConfig(mode=mode, loss_fn=fn, conf_thresh=tau, conf_per_pixel=pp, align_corners=ac)
    ls_d = cu(ls).requires_grad_(True)
    unsup, closs, rate = ops.consistency_loss(ls_d, cu(l0), cu(l1) if mode == 'mix' else None, (H, W), cfg,
                                              ranges=ops.ranges_to_device(ranges, DEV), um0=cu(um0), um1=cu(um1),
                                              cons_weight=0.7)
    unsup.backward()
    assert float(closs) == pytest.approx(float(r['consistency_loss'].detach()), rel=1e-4, abs=1e-9)
    assert float(unsup) == pytest.approx(float(r['unsup_loss'].detach()), rel=1e-4, abs=1e-9)
    if tau > 0:
        assert float(rate) == pytest.approx(float(r['conf_rate']), abs=3.0 / (N * H * W))
    want = ls_o.grad.numpy()
    got = ls_d.grad.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=3e-5 * np.abs(want).max())


def test_consistency_properties_full_size(ops):
    """cfg-3 size: identical student / pasted teacher => zero loss and zero gradient; loss is linear in cons_weight;
    valid masks of zeros kill it; rate == 1 when the threshold is tiny."""
    N, C, h, w, H, W = 4, 19, 65, 129, 512, 1024
    gen = torch.Generator(device=DEV).manual_seed(1)
    l0 = torch.randn(N, C, h, w, generator=gen, device=DEV) * 3
    l1 = torch.randn(N, C, h, w, generator=gen, device=DEV) * 3
    import mask_gen
    ranges = ops.ranges_to_device(mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(
        N, (H, W), rng=np.random.RandomState(0)), DEV)
    cfg = ops.ConsistencyConfig(mode='cut', loss_fn='var', conf_thresh=1e-6, conf_per_pixel=False)
    sc, _ = ops.consistency_forward(cfg, l0, l0, None, (H, W), ranges=ranges)
    assert float(sc[0]) == 0.0 and float(sc[1]) == 1.0
    cfg = ops.ConsistencyConfig(mode='mix', loss_fn='var', conf_thresh=0.0)
    s1, ctx = ops.consistency_forward(cfg, l0, l0, l1, (H, W), ranges=ranges, cons_weight=1.0)
    s3, _ = ops.consistency_forward(cfg, l0, l0, l1, (H, W), ranges=ranges, cons_weight=3.0)
    assert float(s3[3]) == pytest.approx(3.0 * float(s1[3]), rel=1e-6)
    assert float(s3[0]) == pytest.approx(float(s1[0]), rel=1e-7)
    zeros = torch.zeros(N, 1, H, W, device=DEV)
    sz, cz = ops.consistency_forward(cfg, l0, l0, l1, (H, W), ranges=ranges, um0=zeros, um1=zeros)
    assert float(sz[0]) == 0.0
    assert float(ops.consistency_backward(cz, sz).abs().max()) == 0.0
    # gradient only where the pasted teacher differs (box region): outside the box student == teacher
    g = ops.consistency_backward(ctx, s1)
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    # repeated launches are deterministic in the forward
    s1b, _ = ops.consistency_forward(cfg, l0, l0, l1, (H, W), ranges=ranges, cons_weight=1.0)
    assert torch.equal(torch.nan_to_num(s1, nan=-1.0), torch.nan_to_num(s1b, nan=-1.0))


def test_consistency_backward_over_runs_of_samples_equals_the_whole_launch(ops):
    """Round 5: `consistency_backward(..., samples=(s0, s1))` -- the step issues the backward as two halves on two streams. The per-pixel
    work is independent between samples: with the reproducible tile order the two halves equal the whole launch bit for bit."""
    N, C, h, w, H, W = 6, 21, 41, 41, 321, 321
    gen = torch.Generator(device=DEV).manual_seed(7)
    ls, l0, l1 = (torch.randn(N, C, h, w, generator=gen, device=DEV) * 3 for _ in range(3))
    import mask_gen
    ranges = ops.ranges_to_device(mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(
        N, (H, W), rng=np.random.RandomState(3)), DEV)
    um0 = (torch.rand(N, 1, H, W, generator=gen, device=DEV) > 0.1).float()
    cfg = ops.ConsistencyConfig(mode='mix', loss_fn='var', conf_thresh=0.3)
    ops.set_deterministic_wgrad(True)            # (colour-class launches: one add per low-resolution cell and launch)
    try:
        sc, ctx = ops.consistency_forward(cfg, ls, l0, l1, (H, W), ranges=ranges, um0=um0, um1=None)
        whole = ops.consistency_backward(ctx, sc, torch.zeros_like(ls))
        parts = torch.zeros_like(ls)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.consistency_backward(ctx, sc, parts, samples=(0, 2))
        ops.consistency_backward(ctx, sc, parts, samples=(2, N))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
    finally:
        ops.set_deterministic_wgrad(False)
    assert float(whole.abs().max()) > 0 and torch.equal(whole, parts)
    with pytest.raises(ValueError):
        ops.consistency_backward(ctx, sc, parts, samples=(3, 3))


def test_consistency_error_behaviour(ops):
    with pytest.raises(ValueError, match='Unknown consistency loss function'):
        ops.ConsistencyConfig(loss_fn='nope')
    with pytest.raises(ValueError, match='Unknown mask_mode'):
        ops.ConsistencyConfig(mode='nope')
    l = torch.zeros(1, 3, 4, 4, device=DEV)
    cfg = ops.ConsistencyConfig(mode='mix')
    with pytest.raises(ValueError):                      # mix mode without the second teacher tensor
        ops.consistency_forward(cfg, l, l, None, (4, 4), mask=torch.zeros(1, 1, 4, 4, device=DEV))
    with pytest.raises(ValueError):                      # neither ranges nor mask
        ops.consistency_forward(cfg, l, l, l, (4, 4))


# =========================================================================================== supervised CE
@pytest.mark.parametrize('C', [21, 2])
def test_ce_vs_golden(ops, C):
    g = load_golden('losses')
    pre = 'C{}__'.format(C)
    l = cu(g[pre + 'l_stu']).requires_grad_(True)
    for ldt in (torch.uint8, torch.int64):
        l.grad = None
        y = cu(g[pre + 'labels']).to(ldt)
        ce = ops.cross_entropy(l, y)
        ce.backward()
        assert float(ce) == pytest.approx(float(g[pre + 'ce__val']), rel=1e-5)
        np.testing.assert_allclose(l.grad.cpu().numpy(), g[pre + 'ce__grad'], rtol=1e-3, atol=1e-8)


@pytest.mark.parametrize('geo', [dict(N=2, C=21, h=41, w=41, H=321, W=321, ac=True),
                                 dict(N=1, C=19, h=65, w=129, H=512, W=1024, ac=True),
                                 dict(N=2, C=6, h=9, w=11, H=40, W=57, ac=False)])
def test_ce_fused_upsample_vs_oracle(ops, geo):
    from oracle import losses as olosses
    N, C, h, w, H, W, ac = (geo[k] for k in ('N', 'C', 'h', 'w', 'H', 'W', 'ac'))
    gen = torch.Generator().manual_seed(5)
    lo = torch.randn(N, C, h, w, generator=gen) * 2
    y = torch.randint(0, C, (N, H, W), generator=gen)
    y[torch.rand(N, H, W, generator=gen) < 0.05] = 255
    lo_o = lo.clone().requires_grad_(True)
    ce_o = olosses.supervised_ce(olosses.upsample(lo_o, (H, W), ac), y)
    ce_o.backward()
    lo_d = cu(lo).requires_grad_(True)
    ce_d = ops.cross_entropy(lo_d, cu(y).to(torch.uint8), (H, W), 255, ac)
    ce_d.backward()
    assert float(ce_d) == pytest.approx(float(ce_o.detach()), rel=2e-5)
    want = lo_o.grad.numpy()
    np.testing.assert_allclose(lo_d.grad.cpu().numpy(), want, rtol=2e-3, atol=3e-5 * np.abs(want).max())


def test_ce_all_ignored_is_nan_like_torch(ops):
    l = torch.randn(1, 3, 4, 4, device=DEV)
    y = torch.full((1, 4, 4), 255, dtype=torch.uint8, device=DEV)
    sc, _ = ops.ce_forward(l, y)
    assert torch.isnan(sc[0])          # nn.CrossEntropyLoss gives nan when every label is ignored


# =========================================================================================== bilinear upsample
@pytest.mark.parametrize('ac', [True, False])
def test_upsample_vs_golden(ops, ac):
    g = load_golden('losses')
    lo = cu(g['up__lo']).requires_grad_(True)
    hi = ops.upsample_bilinear(lo, (33, 41), align_corners=ac)
    np.testing.assert_allclose(hi.detach().cpu().numpy(), g['up__hi_ac{}'.format(int(ac))], rtol=1e-5, atol=1e-6)
    (hi * cu(g['up__wgt_ac{}'.format(int(ac))])).sum().backward()
    np.testing.assert_allclose(lo.grad.cpu().numpy(), g['up__grad_ac{}'.format(int(ac))], rtol=1e-4, atol=1e-5)


def test_upsample_full_size_properties(ops):
    x = torch.randn(2, 19, 65, 129, device=DEV)
    hi = ops.upsample_bilinear(x, (512, 1024), True)
    # align_corners=True reproduces the corner samples exactly; constant maps stay constant; range is preserved
    assert torch.equal(hi[:, :, 0, 0], x[:, :, 0, 0]) and torch.equal(hi[:, :, -1, -1], x[:, :, -1, -1])
    assert float(hi.max()) <= float(x.max()) + 1e-5 and float(hi.min()) >= float(x.min()) - 1e-5
    ones = ops.upsample_bilinear(torch.ones(1, 1, 65, 129, device=DEV), (512, 1024), True)
    assert float((ones - 1).abs().max()) < 1e-6
    assert torch.equal(ops.upsample_bilinear(x, (65, 129), True), x)
    # adjoint identity <U x, y> == <x, U^T y>
    xg = x.clone().requires_grad_(True)
    y = torch.randn(2, 19, 512, 1024, device=DEV)
    lhs = (ops.upsample_bilinear(xg, (512, 1024), True) * y).sum()
    lhs.backward()
    rhs = (xg.grad * x).sum()
    assert float(lhs) == pytest.approx(float(rhs), rel=1e-4)


# =========================================================================================== EMA / optimizers
@pytest.mark.parametrize('alpha', [0.99, 0.5, 0.999])
def test_ema_bit_exact_vs_golden(alpha):
    import optim_weight_ema
    g = load_golden('ema')
    keys = [str(k) for k in g['keys']]

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(3, 4, 3, bias=True)
            self.bn = torch.nn.BatchNorm2d(4)
            self.fc = torch.nn.Linear(4, 2)
    tag = 'a{}'.format(alpha)
    stu, tea = Toy().to(DEV), Toy().to(DEV)
    with torch.no_grad():
        for k, t in tea.state_dict().items():
            t.copy_(cu(g['{}__init__{}'.format(tag, k)]) if t.dtype == torch.float32 else torch.tensor(26))
        for k, t in stu.state_dict().items():
            if t.dtype == torch.float32:
                t.copy_(cu(g['{}__init__{}'.format(tag, k)]))
    opt = optim_weight_ema.EMAWeightOptimizer(tea, stu, alpha)
    assert list(tea.state_dict().keys()) == keys
    for step in range(3):
        with torch.no_grad():
            for k, t in stu.state_dict().items():
                if t.dtype == torch.float32:
                    t.copy_(cu(g['{}__src{}__{}'.format(tag, step, k)]))
        opt.step()
        for k, t in tea.state_dict().items():
            np.testing.assert_array_equal(t.cpu().numpy(), g['{}__tgt{}__{}'.format(tag, step, k)])


def test_ema_key_mismatch_raises():
    import optim_weight_ema
    a = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3)).to(DEV)
    b = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4)).to(DEV)
    with pytest.raises(ValueError, match='same state dict keys'):
        optim_weight_ema.EMAWeightOptimizer(a, b, 0.9)


def test_ema_full_size_bit_exact(ops):
    from oracle import ema_opt
    n = 44153876                                       # float state of DeepLab v2 / ResNet-101
    g = torch.Generator(device=DEV).manual_seed(0)
    t = torch.randn(n, generator=g, device=DEV)
    s = torch.randn(n, generator=g, device=DEV)
    t0 = t.cpu().numpy()
    ops.ema_flat(t, s, 0.99)
    np.testing.assert_array_equal(t.cpu().numpy(), ema_opt.ema_step(t0, s.cpu().numpy(), 0.99))
    # alpha = 1 is the identity, alpha = 0 copies the source
    t1 = t.clone()
    ops.ema_flat(t1, s, 1.0)
    assert torch.equal(t1, t)
    ops.ema_flat(t1, s, 0.0)
    assert torch.equal(t1, s)


class _OneParam(torch.nn.Module):
    def __init__(self, p0):
        super().__init__()
        self.p = torch.nn.Parameter(torch.tensor(p0))


@pytest.mark.parametrize('name', ['adam', 'sgd', 'sgd_nesterov'])
@pytest.mark.parametrize('k', [1, 3, 4])
def test_fused_optimizers_vs_golden(name, k):
    from cutmix_semisup_seg_amd import optim as fo
    g = load_golden('optim')
    key = '{}__k{}'.format(name, k)
    mod = _OneParam(g[key + '__p0']).to(DEV)
    groups = [dict(params=[mod.p] * k, lr=3e-3)]
    if name == 'adam':
        opt = fo.FusedAdam(mod, groups)
    else:
        opt = fo.FusedSGD(mod, groups, momentum=0.9, nesterov=(name == 'sgd_nesterov'), weight_decay=5e-4)
    assert opt.k_updates['p'] == k
    for s in range(3):
        opt.zero_grad()
        mod.p.grad.copy_(cu(g[key + '__grads'][s]))
        opt.step()
        np.testing.assert_allclose(mod.p.detach().cpu().numpy(), g[key + '__ps'][s], rtol=3e-6, atol=2e-7)
    assert int(opt.step_count) == 3
    if name == 'adam':
        np.testing.assert_allclose(opt.slot0[:257].cpu().numpy(), g[key + '__m'], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(opt.slot1[:257].cpu().numpy(), g[key + '__v'], rtol=1e-5, atol=1e-12)
    else:
        np.testing.assert_allclose(opt.slot0[:257].cpu().numpy(), g[key + '__buf'], rtol=3e-6, atol=2e-7)


def test_fused_optimizer_rejects_param_in_two_groups():
    from cutmix_semisup_seg_amd import optim as fo
    mod = _OneParam(np.zeros(8, np.float32)).to(DEV)
    with pytest.raises(ValueError, match='more than one parameter group'):
        fo.FusedAdam(mod, [dict(params=[mod.p], lr=1e-3), dict(params=[mod.p], lr=1e-4)])


# =========================================================================================== evaluation
@pytest.mark.parametrize('C', [2, 19, 21])
def test_evaluator_iou_integer_exact(C):
    import evaluation
    g = load_golden('evaluation')
    ev, ev2 = evaluation.EvaluatorIoU(C), evaluation.EvaluatorIoU(C)
    for s in range(3):
        ev.sample(g['C{}__truth{}'.format(C, s)], g['C{}__pred{}'.format(C, s)], ignore_value=255)
        ev2.sample(g['C{}__truth_noign{}'.format(C, s)].astype(np.int64), g['C{}__pred{}'.format(C, s)].astype(np.int64))
    np.testing.assert_array_equal(ev.intersection, g['C{}__intersection'.format(C)])
    np.testing.assert_array_equal(ev.union, g['C{}__union'.format(C)])
    np.testing.assert_array_equal(ev.cm, g['C{}__cm'.format(C)])
    np.testing.assert_array_equal(ev.score(), g['C{}__score'.format(C)])
    np.testing.assert_array_equal(ev2.cm, g['C{}__noign_cm'.format(C)])
    np.testing.assert_array_equal(ev2.score(), g['C{}__noign_score'.format(C)])
    i, u, cm = evaluation.per_class_i_and_u_cm(g['C{}__pred0'.format(C)], g['C{}__truth0'.format(C)], C, 255)
    from oracle import evaluation as oe
    oi, ou, ocm = oe.per_class_iu(g['C{}__pred0'.format(C)].astype(np.int64), g['C{}__truth0'.format(C)].astype(np.int64),
                                  C, 255)
    np.testing.assert_array_equal(i, oi)
    np.testing.assert_array_equal(u, ou)
    np.testing.assert_array_equal(cm, ocm)
    np.testing.assert_array_equal(evaluation.fast_cm(g['C{}__truth_noign0'.format(C)], g['C{}__pred0'.format(C)], C),
                                  oe.confusion(g['C{}__truth_noign0'.format(C)], g['C{}__pred0'.format(C)], C))


def test_evaluator_fill_holes_needs_two_classes():
    import evaluation
    with pytest.raises(ValueError, match='num_classes must be 2'):
        evaluation.EvaluatorIoU(3, fill_holes=True)


@pytest.mark.parametrize('geo', [dict(N=2, C=21, h=41, w=41, H=321, W=321), dict(N=4, C=19, h=65, w=129, H=512, W=1024),
                                 dict(N=2, C=5, h=33, w=33, H=33, W=33)])
def test_fused_argmax_confusion(ops, geo):
    import evaluation
    from oracle import evaluation as oe
    N, C, h, w, H, W = (geo[k] for k in ('N', 'C', 'h', 'w', 'H', 'W'))
    gen = torch.Generator().manual_seed(11)
    lo = torch.randn(N, C, h, w, generator=gen) * 3
    y = torch.randint(0, C, (N, H, W), generator=gen)
    y[torch.rand(N, H, W, generator=gen) < 0.05] = 255
    lo_d = cu(lo)
    ev = evaluation.EvaluatorIoU(C)
    ev.sample_logits(lo_d, cu(y).to(torch.uint8), (H, W), ignore_value=255, align_corners=True)
    # the prediction the kernel made, then integer-exact bookkeeping against the oracle on that same prediction
    _, pred = ops.argmax_confusion(lo_d, None, C, (H, W), want_pred=True)
    acc = oe.IoUAccumulator(C)
    for i in range(N):
        acc.sample(y[i].numpy(), pred[i].cpu().numpy().astype(np.int64), 255)
    np.testing.assert_array_equal(ev.cm, acc.cm)
    np.testing.assert_array_equal(ev.score(), acc.score())
    assert int(ev.cm.sum()) == int((y != 255).sum())              # every valid pixel counted exactly once
    # and the prediction itself equals argmax of the (device) upsample except at fp ties
    ref = ops.upsample_bilinear(lo_d, (H, W), True).argmax(dim=1)
    assert float((ref != pred.long()).float().mean()) < 1e-5


# =========================================================================================== network + whole step
def _cf_input(n, h, w, phase):
    idx = torch.arange(n * 3 * h * w, dtype=torch.float64)
    return torch.sin(phase + 0.61803398875 * idx).reshape(n, 3, h, w).float() * 1.5


_DM = load_golden_json('deeplab2_meta')


@pytest.mark.parametrize('tag,shapes', [('tiny', [(2, 33, 33), (1, 40, 57)]), ('r101', [(2, 33, 33), (1, 65, 97)])])
def test_deeplab2_fp32_vs_golden(tag, shapes):
    from architectures import deeplab2
    from oracle import deeplab2 as odl
    g = load_golden('deeplab2')
    meta = _DM[tag]
    C, layers = meta['num_classes'], meta['layers']
    net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    net.load_state_dict(odl.closed_form_state(C, layers))
    net = net.to(DEV)
    net.compute_dtype = torch.float32
    net.train()
    net.freeze_batchnorm()
    for ii, (n, h, w) in enumerate(shapes):
        key = '{}__in{}'.format(tag, ii)
        x = cu(_cf_input(n, h, w, 0.3 + ii))
        net.zero_grad()
        gerr = {}
        assert net._use_hip_body()
        lo = net.forward_lowres(x)
        np.testing.assert_allclose(lo.detach().cpu().numpy(), g[key + '__lowres'], rtol=2e-3, atol=2e-4)
        y = net(x)
        np.testing.assert_allclose(y.detach().cpu().numpy()[:, :, ::4, ::4], g[key + '__full_sub4'], rtol=2e-3, atol=2e-4)
        wsum = torch.cos(0.11 * torch.arange(y.numel(), dtype=torch.float64)).reshape(y.shape).float().to(DEV)
        (y * wsum).sum().backward()
        named = dict(net.named_parameters())
        for k in ['conv1.weight', 'layer1.0.conv2.weight', 'layer3.0.downsample.0.weight', 'layer4.0.conv3.weight',
                  'layer5.conv2d_list.0.weight', 'layer5.conv2d_list.1.bias']:
            want = g['{}__grad__{}'.format(key, k)]
            got = named[k].grad.cpu().numpy().reshape(-1)[:4096]
            # fp32 compute = the PARITY configuration: the body runs on the f32-input MFMA kernels (csrc/conv_f32.hip);
            # only the stem still goes through the library
            np.testing.assert_allclose(got, want, rtol=5e-2, atol=1.5e-2 * (np.abs(want).max() + 1e-12))
            gerr[k] = float(np.abs(got - want).max() / (np.abs(want).max() + 1e-12))
        # ASPP d18 / d24 never receive a gradient: None on the library engine, an untouched all-zero view of
        # the gradient arena on the hand-written engine (the fused optimizer skips them either way, k_updates == 0)
        for k in ('layer5.conv2d_list.2.weight', 'layer5.conv2d_list.3.bias'):
            assert named[k].grad is None or float(named[k].grad.abs().max()) == 0.0
        print('fp32 HIP engine vs golden [{}]: max gradient error / gradient scale per tensor: {}'.format(key, gerr))


def test_deeplab2_bf16_close_to_fp32_oracle():
    from architectures import deeplab2
    from oracle import deeplab2 as odl
    C, layers = 21, [3, 4, 23, 3]
    st = odl.closed_form_state(C, layers)
    st_bf = {k: (v.bfloat16().float() if (v.dtype == torch.float32 and v.dim() == 4) else v) for k, v in st.items()}
    x = _cf_input(1, 65, 97, 1.3).bfloat16().float()
    want = odl.forward_lowres(x, st_bf, layers, frozen=True).numpy()
    net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    net.load_state_dict(st_bf)
    net = net.to(DEV)
    net.train()
    net.freeze_batchnorm()
    with torch.no_grad():
        got = net.forward_lowres(cu(x)).cpu().numpy()
    scale = np.abs(want).max()
    # bf16 activations through 104 convolutions against the fp32 oracle on identical bf16-rounded weights
    assert np.abs(got - want).max() <= 0.12 * scale
    assert np.abs(got - want).mean() <= 2.5e-2 * scale


@pytest.mark.parametrize('cfg_name,cfg', [
    ('adam_var_mix', dict(opt='adam', fn='var', mode='mix', tau=0.3, pp=False)),
    ('sgd_kld_cut_pp', dict(opt='sgd', fn='kld', mode='cut', tau=0.3, pp=True))])
@pytest.mark.parametrize('fuse', [True, False])
def test_whole_step_three_iterations_vs_golden(ops, cfg_name, cfg, fuse):
    """fp32 compute; the reference's modules produced step.npz on the CPU."""
    from architectures import deeplab2
    from oracle import deeplab2 as odl
    import mask_gen
    import optim_weight_ema
    from cutmix_semisup_seg_amd import optim as fo
    from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig, UnsupBatch
    g = load_golden('step')
    C, layers = 5, [1, 1, 1, 1]
    N, H, W = 2, 33, 33
    mk = lambda: deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    stu, tea = mk(), mk()
    stu.load_state_dict(odl.closed_form_state(C, layers))
    stu, tea = stu.to(DEV), tea.to(DEV)
    stu.compute_dtype = tea.compute_dtype = torch.float32
    lr = 1e-3
    groups = [dict(params=list(stu.pretrained_parameters()), lr=lr * 0.1), dict(params=list(stu.new_parameters()), lr=lr)]
    if cfg['opt'] == 'adam':
        opt = fo.FusedAdam(stu, groups)
    else:
        opt = fo.FusedSGD(stu, groups, momentum=0.9, nesterov=False, weight_decay=5e-4)
    for p in tea.parameters():
        p.requires_grad = False
    ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
    ema.fuse_into(opt)
    gen_m = mask_gen.BoxMaskGenerator(0.5, invert=True)
    rng = np.random.RandomState(12345)
    stu.train(); tea.train(); stu.freeze_batchnorm(); tea.freeze_batchnorm()
    scfg = StepConfig(mask_mode='mix' if cfg['mode'] == 'mix' else 'zero', cons_loss_fn=cfg['fn'], conf_thresh=cfg['tau'],
                      conf_per_pixel=cfg['pp'], fuse_batches=fuse, compute_dtype=torch.float32)
    step = CutMixMeanTeacherStep(stu, tea, opt, ema, scfg)
    for it in range(3):
        gen = torch.Generator().manual_seed(1000 + it)
        x = torch.randn(N, 3, H, W, generator=gen)
        y = torch.randint(0, C, (N, 1, H, W), generator=gen)
        y[torch.rand(N, 1, H, W, generator=gen) < 0.05] = 255
        ux0 = torch.randn(N, 3, H, W, generator=gen)
        ux1 = torch.randn(N, 3, H, W, generator=gen)
        ranges = ops.ranges_to_device(gen_m.generate_ranges(N, (H, W), rng=rng), DEV)
        ub = UnsupBatch(cu(ux0), ranges, x1_tea=cu(ux1) if cfg['mode'] == 'mix' else None)
        r = step(cu(x), cu(y).to(torch.uint8), [ub])
        want = g[cfg_name + '__log'][it]
        assert float(r['sup_loss']) == pytest.approx(want[0], rel=2e-3)
        assert float(r['consistency_loss']) == pytest.approx(want[1], rel=2e-2, abs=1e-8)
        assert float(r['conf_rate']) == pytest.approx(want[2], abs=5e-3)
        fs = lambda sd: float(sum((v.double() ** 2).sum() for v in sd.values() if v.dtype == torch.float32))
        assert fs(stu.state_dict()) == pytest.approx(want[4], rel=1e-4)
        assert fs(tea.state_dict()) == pytest.approx(want[6], rel=1e-4)
    for k in ('conv1.weight', 'layer3.0.conv2.weight', 'layer5.conv2d_list.1.weight', 'layer5.conv2d_list.3.weight'):
        np.testing.assert_allclose(stu.state_dict()[k].cpu().numpy().reshape(-1)[:2048],
                                   g['{}__stu__{}'.format(cfg_name, k)], rtol=5e-3, atol=5e-5)
        np.testing.assert_allclose(tea.state_dict()[k].cpu().numpy().reshape(-1)[:2048],
                                   g['{}__tea__{}'.format(cfg_name, k)], rtol=5e-3, atol=5e-6)
    assert not step.nan_detected()


@pytest.mark.parametrize('fuse', [True, False])
def test_pi_model_step_teacher_is_the_student(ops, fuse):
    """`--model pi` (train_seg_semisup_mask_mt.py:110-113): the teacher IS the student, there is no EMA. In the first
    iteration a mean teacher starts as a copy of the student, so losses and the updated student must coincide with the
    mean-teacher step on the same data; the step must neither touch a second network nor fork a stream race on one."""
    from architectures import deeplab2
    from oracle import deeplab2 as odl
    import mask_gen
    import optim_weight_ema
    from cutmix_semisup_seg_amd import optim as fo
    from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig, UnsupBatch
    C, layers, N, H, W, lr = 5, [1, 1, 1, 1], 2, 33, 33, 1e-3

    def build(pi):
        stu = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
        stu.load_state_dict(odl.closed_form_state(C, layers))
        stu = stu.to(DEV)
        stu.compute_dtype = torch.float32
        opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=lr * 0.1),
                                 dict(params=list(stu.new_parameters()), lr=lr)])
        if pi:
            tea, ema = stu, None
        else:
            tea = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3)).to(DEV)
            tea.compute_dtype = torch.float32
            for p in tea.parameters():
                p.requires_grad = False
            ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
            ema.fuse_into(opt)
        stu.train(); tea.train(); stu.freeze_batchnorm(); tea.freeze_batchnorm()
        cfg = StepConfig(mask_mode='mix', cons_loss_fn='var', conf_thresh=0.3, fuse_batches=fuse,
                         compute_dtype=torch.float32)
        return stu, CutMixMeanTeacherStep(stu, tea, opt, ema, cfg)

    gen = torch.Generator().manual_seed(5)
    x = torch.randn(N, 3, H, W, generator=gen)
    y = torch.randint(0, C, (N, 1, H, W), generator=gen)
    ux0, ux1 = torch.randn(N, 3, H, W, generator=gen), torch.randn(N, 3, H, W, generator=gen)
    rn = mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(N, (H, W), rng=np.random.RandomState(3))
    res, weights = [], []
    for pi in (True, False):
        stu, step = build(pi)
        ub = UnsupBatch(cu(ux0), ops.ranges_to_device(rn, DEV), x1_tea=cu(ux1))
        r = step(cu(x), cu(y).to(torch.uint8), [ub])
        res.append([float(r[k]) for k in ('sup_loss', 'consistency_loss', 'conf_rate')])
        weights.append({k: v.clone() for k, v in stu.state_dict().items() if v.dtype == torch.float32})
    assert res[0] == pytest.approx(res[1], rel=1e-4, abs=1e-6)
    for k in weights[0]:
        torch.testing.assert_close(weights[0][k], weights[1][k], rtol=1e-3, atol=2e-5)     # (library conv run-to-run noise through the first Adam update: lr * g / (|g| + eps))


@pytest.mark.parametrize('fuse', [True, False])
def test_pi_model_step_vs_oracle(ops, fuse):
    """`--model pi` against the CPU oracle (oracle/step.py, pi_model=True): two iterations, losses, confidence rate and the
    updated student (fp32 parity configuration of the hand-written engine)."""
    from architectures import deeplab2
    from oracle import deeplab2 as odl, step as ostep, boxmask as obox
    import mask_gen
    from cutmix_semisup_seg_amd import optim as fo
    from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig, UnsupBatch
    C, layers, N, H, W, lr = 5, [1, 1, 1, 1], 2, 33, 33, 1e-3
    st = odl.closed_form_state(C, layers)
    stu = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    stu.load_state_dict(st)
    stu = stu.to(DEV)
    stu.compute_dtype = torch.float32
    stu.engine_kind = 'hip'
    opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=lr * 0.1),
                             dict(params=list(stu.new_parameters()), lr=lr)])
    stu.train(); stu.freeze_batchnorm()
    step = CutMixMeanTeacherStep(stu, stu, opt, None, StepConfig(conf_thresh=0.3, fuse_batches=fuse, compute_dtype=torch.float32))
    S = ostep.StepState(st, C, layers, opt='adam', lr=lr)
    gen_m = mask_gen.BoxMaskGenerator(0.5, invert=True)
    rng = np.random.RandomState(11)
    ones = torch.ones(N, 1, H, W)
    for it in range(2):
        g = torch.Generator().manual_seed(300 + it)
        x = torch.randn(N, 3, H, W, generator=g)
        y = torch.randint(0, C, (N, 1, H, W), generator=g)
        y[torch.rand(N, 1, H, W, generator=g) < 0.05] = 255
        ux0, ux1 = torch.randn(N, 3, H, W, generator=g), torch.randn(N, 3, H, W, generator=g)
        rn = gen_m.generate_ranges(N, (H, W), rng=rng)
        r = step(cu(x), cu(y).to(torch.uint8), [UnsupBatch(cu(ux0), ops.ranges_to_device(rn, DEV), x1_tea=cu(ux1))])
        m = torch.tensor(obox.rasterise(rn, (H, W), True).astype(np.float32))
        want = ostep.train_iteration(S, x, y, ux0, ux1, ones, ones, m, conf_thresh=0.3, pi_model=True)
        assert float(r['sup_loss']) == pytest.approx(want['sup_loss'], rel=2e-4)
        assert float(r['consistency_loss']) == pytest.approx(want['consistency_loss'], rel=2e-3, abs=1e-9)
        assert float(r['conf_rate']) == pytest.approx(want['conf_rate'], abs=2e-3)
    for k in ('conv1.weight', 'layer3.0.conv2.weight', 'layer5.conv2d_list.1.weight'):
        got, want_w = stu.state_dict()[k].cpu(), S.student[k]
        bad = ~torch.isclose(got, want_w, rtol=5e-3, atol=5e-5)
        # Adam's first updates are lr * sign-like: an element whose gradient is ~0 may step the other way (measured: 1 of
        # 9408 stem weights off by one lr-sized step); everything else agrees
        assert float(bad.float().mean()) <= 1e-3 and float((got - want_w).abs().max()) <= 2.5 * lr, k


def test_loss_backward_one_launch_equals_colour_classes(ops):
    """cms_loss_set_deterministic: the tiled backward kernels as ONE launch (default: fp32 atomics between tiles that share a
    low-resolution cell, run-dependent order) give the gradient of the colour-class launches (run-to-run reproducible) up to the
    order of a handful of fp32 additions; the colour-class mode reproduces itself bit for bit."""
    from cutmix_semisup_seg_amd._lib import fn
    N, C, h, w, H, W = 4, 21, 41, 41, 321, 321
    gen = torch.Generator(device=DEV).manual_seed(3)
    ls = torch.randn(N, C, h, w, generator=gen, device=DEV) * 2
    l0 = torch.randn(N, C, h, w, generator=gen, device=DEV) * 2
    l1 = torch.randn(N, C, h, w, generator=gen, device=DEV) * 2
    y = torch.randint(0, C, (N, 1, H, W), generator=gen, device=DEV).to(torch.uint8)
    import mask_gen
    ranges = ops.ranges_to_device(mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(N, (H, W), rng=np.random.RandomState(2)), DEV)
    cfg = ops.ConsistencyConfig(mode='mix', loss_fn='var', conf_thresh=0.0)

    def grads():
        sc, ctx = ops.consistency_forward(cfg, ls, l0, l1, (H, W), ranges=ranges)
        g1 = ops.consistency_backward(ctx, sc)
        cs, cctx = ops.ce_forward(ls, y, (H, W), 255, True)
        g2 = ops.ce_backward(cctx, cs)
        torch.cuda.synchronize()
        return g1.clone(), g2.clone()
    try:
        fn['cms_loss_set_deterministic'](1)
        a1, a2 = grads()
        b1, b2 = grads()
        assert torch.equal(a1, b1) and torch.equal(a2, b2)
        fn['cms_loss_set_deterministic'](0)
        c1, c2 = grads()
    finally:
        fn['cms_loss_set_deterministic'](1 if ops.deterministic_wgrad() else 0)
    assert float((c1 - a1).abs().max()) <= 1e-6 * float(a1.abs().max())
    assert float((c2 - a2).abs().max()) <= 1e-6 * float(a2.abs().max())


# ---------------------------------------------------------------------------------------------------------------------------
# Round 6: each loss as ONE launch (forward + backward; the gradient's scalar factor applied afterwards)
@pytest.mark.parametrize('geo', [
    dict(N=2, C=5, h=6, w=7, H=41, W=50, ac=True),
    dict(N=2, C=21, h=41, w=41, H=321, W=321, ac=True),       # cfg 2 geometry (Pascal crop)
    dict(N=1, C=19, h=65, w=129, H=512, W=1024, ac=True),     # cfg 3 geometry (Cityscapes)
    dict(N=2, C=7, h=9, w=9, H=33, W=33, ac=False),           # generic class count, align_corners=False
])
@pytest.mark.parametrize('fn,mode,tau,pp', [('var', 'mix', 0.5, False), ('var', 'mix', 0.6, True),
                                            ('kld', 'cut', 0.5, True), ('bce', 'mix', 0.0, False),
                                            ('logits_var', 'cut', 0.0, False), ('logits_smoothl1', 'mix', 0.4, True)])
def test_consistency_one_launch_vs_oracle_and_vs_the_launch_pair(ops, geo, fn, mode, tau, pp):
    """ops.consistency_fused (cms_consistency_fwd_bwd + finalize + deferred rate) against the oracle's autograd (the bounds of
    test_consistency_fused_upsample_vs_oracle) and against the forward / backward launch pair: same scalars to 1e-6 (the partial
    sums meet in another order), same gradient to 2e-6 of its scale (rate applied after the adjoint sums instead of before)."""
    from oracle import boxmask, losses as olosses
    from cutmix_semisup_seg_amd._lib import fn as cfn
    N, C, h, w, H, W, ac = (geo[k] for k in ('N', 'C', 'h', 'w', 'H', 'W', 'ac'))
    gen = torch.Generator().manual_seed(C * H + w + 1)
    ls = torch.randn(N, C, h, w, generator=gen) * 2
    l0 = torch.randn(N, C, h, w, generator=gen) * 3
    l1 = torch.randn(N, C, h, w, generator=gen) * 3
    um0 = (torch.rand(N, 1, H, W, generator=gen) > 0.2).float()
    um1 = (torch.rand(N, 1, H, W, generator=gen) > 0.2).float()
    ranges = boxmask.rects_to_ranges(boxmask.draw_rects(N, (H, W), 0.5, rng=np.random.RandomState(5)), (H, W))
    m = torch.tensor(boxmask.rasterise(ranges, (H, W), True).astype(np.float32))
    ls_o = ls.clone().requires_grad_(True)
    up = lambda t: olosses.upsample(t, (H, W), align_corners=ac)
    kw = dict(loss_fn=fn, conf_thresh=tau, conf_per_pixel=pp, cons_weight=0.7)
    r = (olosses.mix_mode_loss(up(ls_o), up(l0), up(l1), m, um0, um1, **kw) if mode == 'mix'
         else olosses.cut_mode_loss(up(ls_o), up(l0), m, um0, **kw))
    r['unsup_loss'].backward()
    cfg = ops.ConsistencyConfig(mode=mode, loss_fn=fn, conf_thresh=tau, conf_per_pixel=pp, align_corners=ac)
    args = (cfg, cu(ls), cu(l0), cu(l1) if mode == 'mix' else None, (H, W))
    kwd = dict(ranges=ops.ranges_to_device(ranges, DEV), um0=cu(um0), um1=cu(um1), ramp_val=0.9, cons_weight=0.7)
    d = ops._cons_desc(cfg, args[1], args[2], args[3], kwd['ranges'], None, kwd['um0'], kwd['um1'], (H, W))
    import ctypes
    assert cfn['cms_consistency_fused_supported'](ctypes.byref(d)) == 1
    g_one = torch.zeros(N, C, h, w, device=DEV)
    sc_one = ops.consistency_fused(*args, g_one, **kwd)
    sc_two, ctx = ops.consistency_forward(*args, **kwd)
    g_two = ops.consistency_backward(ctx, sc_two)
    torch.cuda.synchronize()
    a, b = sc_one.cpu().numpy(), sc_two.cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-12, equal_nan=True)
    scale = float(g_two.abs().max())
    assert float((g_one - g_two).abs().max()) <= 2e-6 * scale + 1e-12, (float((g_one - g_two).abs().max()), scale)
    # oracle (ramp 1.0 there: the device ran with ramp_val 0.9, a plain factor of loss and gradient)
    assert float(sc_one[3]) == pytest.approx(0.9 * float(r['unsup_loss'].detach()), rel=1e-4, abs=1e-9)
    if tau > 0:
        assert float(sc_one[1]) == pytest.approx(float(r['conf_rate']), abs=3.0 / (N * H * W))
    want = 0.9 * ls_o.grad.numpy()
    np.testing.assert_allclose(g_one.cpu().numpy(), want, rtol=2e-3, atol=3e-5 * np.abs(want).max())


@pytest.mark.parametrize('geo', [dict(N=2, C=21, h=41, w=41, H=321, W=321, ac=True),
                                 dict(N=1, C=19, h=65, w=129, H=512, W=1024, ac=True),
                                 dict(N=2, C=6, h=9, w=11, H=40, W=57, ac=False)])
def test_ce_one_launch_vs_oracle_and_vs_the_launch_pair(ops, geo):
    from oracle import losses as olosses
    N, C, h, w, H, W, ac = (geo[k] for k in ('N', 'C', 'h', 'w', 'H', 'W', 'ac'))
    gen = torch.Generator().manual_seed(6)
    lo = torch.randn(N, C, h, w, generator=gen) * 2
    y = torch.randint(0, C, (N, H, W), generator=gen)
    y[torch.rand(N, H, W, generator=gen) < 0.05] = 255
    lo_o = lo.clone().requires_grad_(True)
    ce_o = olosses.supervised_ce(olosses.upsample(lo_o, (H, W), ac), y)
    ce_o.backward()
    yd = cu(y).to(torch.uint8)
    g_one = torch.zeros(N, C, h, w, device=DEV)
    sc_one = ops.ce_fused(cu(lo), yd, g_one, (H, W), 255, ac)
    sc_two, ctx = ops.ce_forward(cu(lo), yd, (H, W), 255, ac)
    g_two = ops.ce_backward(ctx, sc_two)
    torch.cuda.synchronize()
    np.testing.assert_allclose(sc_one.cpu().numpy(), sc_two.cpu().numpy(), rtol=2e-6)
    assert float((g_one - g_two).abs().max()) <= 2e-6 * float(g_two.abs().max())
    assert float(sc_one[0]) == pytest.approx(float(ce_o.detach()), rel=2e-5)
    want = lo_o.grad.numpy()
    np.testing.assert_allclose(g_one.cpu().numpy(), want, rtol=2e-3, atol=3e-5 * np.abs(want).max())


def test_one_launch_losses_fall_back_where_the_launch_does_not_exist(ops):
    """Identity geometry (full-resolution logits) and the deterministic mode have no fused launch: the wrappers run the launch pair
    and give its result exactly (identity) / bit-reproducibly (deterministic)."""
    from cutmix_semisup_seg_amd._lib import fn as cfn
    import mask_gen
    N, C, H, W = 2, 5, 33, 47
    gen = torch.Generator(device=DEV).manual_seed(9)
    ls, l0, l1 = (torch.randn(N, C, H, W, generator=gen, device=DEV) for _ in range(3))
    ranges = ops.ranges_to_device(mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(N, (H, W), rng=np.random.RandomState(2)), DEV)
    cfg = ops.ConsistencyConfig(mode='mix', loss_fn='var', conf_thresh=0.3)
    g1 = torch.zeros_like(ls)
    sc1 = ops.consistency_fused(cfg, ls, l0, l1, (H, W), g1, ranges=ranges)
    sc2, ctx = ops.consistency_forward(cfg, ls, l0, l1, (H, W), ranges=ranges)
    g2 = ops.consistency_backward(ctx, sc2)
    assert torch.equal(sc1, sc2) and torch.equal(g1, g2)
    # deterministic mode at an upsampling geometry
    h, w, HH, WW = 9, 9, 65, 65
    ls, l0, l1 = (torch.randn(N, C, h, w, generator=gen, device=DEV) for _ in range(3))
    ranges = ops.ranges_to_device(mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(N, (HH, WW), rng=np.random.RandomState(3)), DEV)
    y = torch.randint(0, C, (N, HH, WW), generator=gen, device=DEV).to(torch.uint8)
    try:
        cfn['cms_loss_set_deterministic'](1)
        outs = []
        for _ in range(2):
            ga, gb = torch.zeros_like(ls), torch.zeros_like(ls)
            sa = ops.consistency_fused(cfg, ls, l0, l1, (HH, WW), ga, ranges=ranges)
            sb = ops.ce_fused(ls, y, gb, (HH, WW), 255, True)
            torch.cuda.synchronize()
            outs.append((sa.clone(), ga.clone(), sb.clone(), gb.clone()))
        assert all(torch.equal(p, q) for p, q in zip(*outs))
    finally:
        cfn['cms_loss_set_deterministic'](1 if ops.deterministic_wgrad() else 0)
