"""
Pins the CPU oracle (oracle/*.py) to the golden vectors produced by the reference's own modules
(tests/golden/make_golden.py). CPU only.
"""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden, load_golden_json
from oracle import boxmask, evaluation as oeval, ema_opt, losses as olosses, deeplab2 as odl, step as ostep


# ---------------------------------------------------------------- box masks: bit exact
_BM = load_golden_json('boxmask_meta')


@pytest.mark.parametrize('case', _BM['cases'], ids=[c['key'] for c in _BM['cases']])
def test_boxmask_bit_exact(case):
    g = load_golden('boxmask')
    kw = dict(_BM['flagsets'][case['flagset']])
    pr = kw.pop('prop_range')
    pr = tuple(pr) if isinstance(pr, list) else pr
    shape = tuple(case['shape'])
    with np.errstate(all='ignore'):
        m = boxmask.generate_params(_BM['n'], shape, pr, rng=np.random.RandomState(case['seed']), **kw)
    assert m.dtype == np.float64 and m.shape == (_BM['n'], 1) + shape
    m8 = m.astype(np.uint8)
    assert hashlib.sha256(m8.tobytes()).hexdigest() == case['sha256']
    assert int(m8.sum()) == case['total']
    np.testing.assert_array_equal(m8.sum(axis=3)[:, 0], g[case['key'] + '__rowsum'])
    np.testing.assert_array_equal(m8.sum(axis=2)[:, 0], g[case['key'] + '__colsum'])
    if case['key'] + '__mask' in g:
        np.testing.assert_array_equal(np.packbits(m8.reshape(_BM['n'], -1), axis=1), g[case['key'] + '__mask'])


# ---------------------------------------------------------------- EMA: bit exact fp32
@pytest.mark.parametrize('alpha', [0.99, 0.5, 0.999])
def test_ema_bit_exact(alpha):
    g = load_golden('ema')
    keys = [str(k) for k in g['keys']]
    tag = 'a{}'.format(alpha)
    tgt = {k: g['{}__init__{}'.format(tag, k)] for k in keys}
    for step in range(3):
        for k in keys:
            src = g['{}__src{}__{}'.format(tag, step, k)]
            want = g['{}__tgt{}__{}'.format(tag, step, k)]
            if src.dtype == np.float32:
                tgt[k] = ema_opt.ema_step(tgt[k], src, alpha)
                np.testing.assert_array_equal(tgt[k], want)
            else:
                # int64 num_batches_tracked is never touched by the EMA (Q5)
                np.testing.assert_array_equal(tgt[k], want)


def test_ema_init_copies_floats_only():
    g = load_golden('ema')
    keys = [str(k) for k in g['keys']]
    # after construction the float tensors equal the student's fill (phase 0.1); the int one keeps the teacher's
    assert g['a0.99__init__bn.num_batches_tracked'] == 26      # int(2.3*10)+3, teacher's own value


# ---------------------------------------------------------------- evaluation: integer exact
@pytest.mark.parametrize('C', [2, 19, 21])
def test_evaluation_exact(C):
    g = load_golden('evaluation')
    acc, acc2 = oeval.IoUAccumulator(C), oeval.IoUAccumulator(C)
    for s in range(3):
        t = g['C{}__truth{}'.format(C, s)].astype(np.int64)
        p = g['C{}__pred{}'.format(C, s)].astype(np.int64)
        acc.sample(t, p, ignore_value=255)
        acc2.sample(g['C{}__truth_noign{}'.format(C, s)].astype(np.int64), p)
        i, u, cm = oeval.per_class_iu(p, t, C, 255)
        i2, u2 = oeval.iu_from_confusion(cm)
        np.testing.assert_array_equal(i, i2)
        np.testing.assert_array_equal(u, u2)
    np.testing.assert_array_equal(acc.intersection, g['C{}__intersection'.format(C)])
    np.testing.assert_array_equal(acc.union, g['C{}__union'.format(C)])
    np.testing.assert_array_equal(acc.cm, g['C{}__cm'.format(C)])
    np.testing.assert_array_equal(acc.score(), g['C{}__score'.format(C)])
    np.testing.assert_array_equal(acc2.intersection, g['C{}__noign_intersection'.format(C)])
    np.testing.assert_array_equal(acc2.cm, g['C{}__noign_cm'.format(C)])
    np.testing.assert_array_equal(acc2.score(), g['C{}__noign_score'.format(C)])


# ---------------------------------------------------------------- LR schedules / rampup
def test_lr_closed_forms():
    g = load_golden('lr')
    base = float(g['base_lr'])
    for i in range(g['poly'].shape[0]):
        assert g['poly'][i, 0] == pytest.approx(ema_opt.poly_lr(base * 0.1, i, 40), rel=1e-12)
        assert g['poly'][i, 1] == pytest.approx(ema_opt.poly_lr(base, i, 40), rel=1e-12)
        assert g['cosine'][i, 1] == pytest.approx(ema_opt.cosine_lr(base, i, 40), rel=1e-9, abs=1e-20)
    for e in range(g['stepped'].shape[0]):
        assert g['stepped'][e, 1] == pytest.approx(ema_opt.multistep_lr(base, e, [3, 6], 0.1), rel=1e-9)
    for e, R, v in g['rampup']:
        assert v == pytest.approx(ema_opt.sigmoid_rampup(e, int(R)), rel=1e-12)


# ---------------------------------------------------------------- optimizers with duplicated entries
@pytest.mark.parametrize('k', [1, 3, 4])
def test_adam_k_fold(k):
    g = load_golden('optim')
    key = 'adam__k{}'.format(k)
    p = g[key + '__p0']
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    step = 0
    for s in range(3):
        p, m, v, step = ema_opt.adam_k_updates(p, g[key + '__grads'][s], m, v, step, 3e-3, k=k)
        np.testing.assert_allclose(p, g[key + '__ps'][s], rtol=2e-6, atol=1e-7)
    assert step == int(g[key + '__step']) == 3 * k
    np.testing.assert_allclose(m, g[key + '__m'], rtol=1e-5, atol=1e-8)    # torch's CPU lerp is FMA-contracted
    np.testing.assert_allclose(v, g[key + '__v'], rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize('name,nesterov', [('sgd', False), ('sgd_nesterov', True)])
@pytest.mark.parametrize('k', [1, 3, 4])
def test_sgd_k_fold(name, nesterov, k):
    g = load_golden('optim')
    key = '{}__k{}'.format(name, k)
    p = g[key + '__p0']
    buf = None
    for s in range(3):
        p, buf = ema_opt.sgd_k_updates(p, g[key + '__grads'][s], buf, 3e-3, k=k, momentum=0.9, nesterov=nesterov,
                                       weight_decay=5e-4)
        np.testing.assert_allclose(p, g[key + '__ps'][s], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(buf, g[key + '__buf'], rtol=2e-6, atol=1e-7)


# ---------------------------------------------------------------- losses
_LC = load_golden_json('losses_meta')


@pytest.mark.parametrize('case', _LC, ids=[c['key'] for c in _LC])
def test_consistency_losses(case):
    g = load_golden('losses')
    pre = 'C{}__'.format(case['C'])
    t = lambda n: torch.tensor(g[pre + n])
    l_stu = t('l_stu').requires_grad_(True)
    kw = dict(loss_fn=case['fn'], conf_thresh=case['conf_thresh'], conf_per_pixel=case['conf_per_pixel'],
              ramp_val=case['ramp_val'], rampup=case['rampup'], cons_weight=case['cons_weight'])
    if case['mode'] == 'mix':
        r = olosses.mix_mode_loss(l_stu, t('l0_tea'), t('l1_tea'), t('mask'), t('um0'), t('um1'), **kw)
    else:
        r = olosses.cut_mode_loss(l_stu, t('l0_tea'), t('mask'), t('um0'), **kw)
    r['unsup_loss'].backward()
    closs, unsup, rate = g[case['key'] + '__vals']
    assert float(r['consistency_loss']) == pytest.approx(closs, rel=2e-6, abs=1e-9)
    assert float(r['unsup_loss']) == pytest.approx(unsup, rel=2e-6, abs=1e-9)
    if case['conf_thresh'] > 0:
        assert float(r['conf_rate']) == pytest.approx(rate, abs=1e-7)
    np.testing.assert_allclose(l_stu.grad.numpy(), g[case['key'] + '__grad'], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('C', [21, 2])
def test_supervised_ce(C):
    g = load_golden('losses')
    pre = 'C{}__'.format(C)
    l = torch.tensor(g[pre + 'l_stu']).requires_grad_(True)
    y = torch.tensor(g[pre + 'labels'].astype(np.int64))
    ce = olosses.supervised_ce(l, y)
    ce.backward()
    assert float(ce) == pytest.approx(float(g[pre + 'ce__val']), rel=2e-6)
    np.testing.assert_allclose(l.grad.numpy(), g[pre + 'ce__grad'], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('ac', [True, False])
def test_upsample(ac):
    g = load_golden('losses')
    lo = torch.tensor(g['up__lo'])
    hi = olosses.upsample(lo, (33, 41), align_corners=ac)
    np.testing.assert_array_equal(hi.numpy(), g['up__hi_ac{}'.format(int(ac))])


# ---------------------------------------------------------------- DeepLab v2
_DM = load_golden_json('deeplab2_meta')


def _cf_input(n, h, w, phase):
    idx = torch.arange(n * 3 * h * w, dtype=torch.float64)
    return torch.sin(phase + 0.61803398875 * idx).reshape(n, 3, h, w).float() * 1.5


@pytest.mark.parametrize('tag,shapes', [('tiny', [(2, 33, 33), (1, 40, 57)]), ('r101', [(2, 33, 33), (1, 65, 97)])])
def test_deeplab2_forward_backward(tag, shapes):
    g = load_golden('deeplab2')
    meta = _DM[tag]
    C, layers = meta['num_classes'], meta['layers']
    st = odl.closed_form_state(C, layers)
    assert len(st) == meta['n_state']
    assert sum(1 for v in st.values() if v.dtype == torch.float32) == meta['n_float']
    assert sum(v.numel() for v in st.values() if v.dtype == torch.float32) == meta['float_elems']
    tk = odl.trainable_keys(C, layers)
    assert sum(st[k].numel() for k in tk) == meta['n_trainable']
    for ii, (n, h, w) in enumerate(shapes):
        x = _cf_input(n, h, w, 0.3 + ii)
        leaves = {k: st[k].clone().requires_grad_(True) for k in tk}
        s2 = dict(st)
        s2.update(leaves)
        taps = {}
        lo = odl.forward_lowres(x, s2, layers, frozen=True, taps=taps)
        y = torch.nn.functional.interpolate(lo, size=(h, w), mode='bilinear', align_corners=True)
        key = '{}__in{}'.format(tag, ii)
        np.testing.assert_allclose(lo.detach().numpy(), g[key + '__lowres'], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(y.detach().numpy()[:, :, ::4, ::4], g[key + '__full_sub4'], rtol=2e-4, atol=2e-5)
        l4 = taps['layer4'].double()
        np.testing.assert_allclose([float(l4.sum()), float((l4 ** 2).sum())], g[key + '__l4_stats'], rtol=1e-4)
        wsum = torch.cos(0.11 * torch.arange(y.numel(), dtype=torch.float64)).reshape(y.shape).float()
        (y * wsum).sum().backward()
        for k in ['conv1.weight', 'layer1.0.conv2.weight', 'layer3.0.downsample.0.weight',
                  'layer4.0.conv3.weight', 'layer5.conv2d_list.0.weight', 'layer5.conv2d_list.1.bias']:
            gr = leaves[k].grad
            want = g['{}__grad__{}'.format(key, k)]
            scale = float(np.abs(want).max()) + 1e-12
            np.testing.assert_allclose(gr.numpy().reshape(-1)[:4096], want, rtol=5e-3, atol=2e-4 * scale)
        none = sorted(k for k in tk if leaves[k].grad is None)
        assert none == meta['none_grads_in{}'.format(ii)]       # ASPP d18/d24 never get gradients (Q1)


def test_deeplab2_batchstat_bn():
    g = load_golden('deeplab2')
    st = odl.closed_form_state(5, [1, 1, 1, 1])
    x = _cf_input(2, 33, 33, 0.3)
    new = {}
    y = odl.forward(x, st, [1, 1, 1, 1], frozen=False, new_stats=new)
    np.testing.assert_allclose(y.numpy()[:, :, ::4, ::4], g['tiny__bnstat__full_sub4'], rtol=2e-4, atol=2e-5)
    for k in ('bn1.running_mean', 'bn1.running_var', 'layer4.0.bn3.running_var',
              'layer2.0.downsample.1.running_mean'):
        np.testing.assert_allclose(new[k].numpy(), g['tiny__bnstat__' + k], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('tag', ['tiny', 'r101'])
def test_param_group_multiplicity(tag):
    meta = _DM[tag]
    C, layers = meta['num_classes'], meta['layers']
    g0, g1 = odl.param_multiplicity(C, layers)
    from collections import Counter
    cnt = Counter(meta['pretrained_order'])
    assert dict(cnt) == dict(g0)
    assert meta['new_order'] == list(g1.keys())
    assert odl.pretrained_param_order(C, layers) == meta['pretrained_order']
    if tag == 'r101':
        hist = Counter(g0.values())
        assert hist == {1: 1, 3: 99, 4: 4}
        assert len(meta['pretrained_order']) == 314


# ---------------------------------------------------------------- whole step, three iterations
@pytest.mark.parametrize('cfg_name,cfg', [
    ('adam_var_mix', dict(opt='adam', fn='var', mode='mix', tau=0.3, pp=False)),
    ('sgd_kld_cut_pp', dict(opt='sgd', fn='kld', mode='cut', tau=0.3, pp=True))])
def test_whole_step(cfg_name, cfg):
    from oracle import boxmask as bm
    g = load_golden('step')
    C, layers = 5, [1, 1, 1, 1]
    N, H, W = 2, 33, 33
    S = ostep.StepState(odl.closed_form_state(C, layers), C, layers, opt=cfg['opt'], lr=1e-3, teacher_alpha=0.99)
    rng = np.random.RandomState(12345)
    for it in range(3):
        gen = torch.Generator().manual_seed(1000 + it)
        x = torch.randn(N, 3, H, W, generator=gen)
        y = torch.randint(0, C, (N, 1, H, W), generator=gen)
        y[torch.rand(N, 1, H, W, generator=gen) < 0.05] = 255
        ux0 = torch.randn(N, 3, H, W, generator=gen)
        ux1 = torch.randn(N, 3, H, W, generator=gen)
        ones = torch.ones(N, 1, H, W)
        m = torch.tensor(bm.generate_params(N, (H, W), 0.5, invert=True, rng=rng).astype(np.float32))
        r = ostep.train_iteration(S, x, y, ux0, ux1, ones, ones, m, mode=cfg['mode'], loss_fn=cfg['fn'],
                                  conf_thresh=cfg['tau'], conf_per_pixel=cfg['pp'])
        want = g[cfg_name + '__log'][it]
        assert r['sup_loss'] == pytest.approx(want[0], rel=1e-4)
        assert r['consistency_loss'] == pytest.approx(want[1], rel=2e-3, abs=1e-9)
        assert r['conf_rate'] == pytest.approx(want[2], abs=2e-3)
        fs = lambda sd: (float(sum(v.double().sum() for v in sd.values() if v.dtype == torch.float32)),
                         float(sum((v.double() ** 2).sum() for v in sd.values() if v.dtype == torch.float32)))
        s1, s2 = fs(S.student)
        t1, t2 = fs(S.teacher)
        assert s2 == pytest.approx(want[4], rel=1e-5)
        assert t2 == pytest.approx(want[6], rel=1e-5)
    for k in ('conv1.weight', 'layer3.0.conv2.weight', 'layer5.conv2d_list.1.weight', 'layer5.conv2d_list.3.weight'):
        np.testing.assert_allclose(S.student[k].numpy().reshape(-1)[:2048], g['{}__stu__{}'.format(cfg_name, k)],
                                   rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(S.teacher[k].numpy().reshape(-1)[:2048], g['{}__tea__{}'.format(cfg_name, k)],
                                   rtol=2e-3, atol=2e-6)
    if cfg['opt'] == 'adam':
        steps = [S.steps[k] for k in ('conv1.weight', 'layer1.0.conv1.weight', 'layer1.0.downsample.0.weight',
                                      'layer5.conv2d_list.0.weight')]
        assert steps == [int(v) for v in g[cfg_name + '__adam_steps']] == [3, 9, 12, 3]


# ---------------------------------------------------------------------------------------------------------- VAT (round 4)
_VM = load_golden_json('vat_meta')


@pytest.mark.parametrize('case', _VM['cases'], ids=[c['key'] for c in _VM['cases']])
def test_vat_direction_and_perturbation_vs_the_reference_closures(case):
    """oracle/vat.py against outputs of the REFERENCE'S OWN closures (t_dot / normalize_eps / normalized_noise_like /
    vat_direction / vat_perburbation, train_seg_semisup_vat_mt.py:213-301), which tests/golden/make_golden.py::gen_vat cuts out
    of the trainer with `ast` and runs on a tiny reference DeepLab v2 with closed-form weights: the initial noise the
    reference drew is part of the fixture, so direction (unit norm per sample), radius (fixed and adaptive) and the teacher
    logits are compared value by value, for all four consistency functions."""
    from oracle import vat as ov
    g = load_golden('vat')
    C, layers = _VM['num_classes'], _VM['layers']
    st = odl.closed_form_state(C, layers)
    x, x_hat = torch.from_numpy(g['x']), torch.from_numpy(g['x_hat'])
    key = case['key']
    eps0 = torch.from_numpy(g[key + '__eps0'])
    # the reference's draw is already normalised and scaled (:222-226, 240-241)
    np.testing.assert_allclose(eps0.reshape(2, -1).norm(dim=1).numpy(), [ov.noise_scale(x.shape)] * 2, rtol=1e-5)
    fnet = lambda t: odl.forward(t, st, layers, frozen=True)
    d, y_logits = ov.vat_direction(fnet, x, x_hat, eps0, case['loss_fn'])
    want_d = torch.from_numpy(g[key + '__direction'])
    cos = (d.reshape(2, -1) * want_d.reshape(2, -1)).sum(dim=1)
    assert float(cos.min()) >= 1.0 - 1e-5, cos
    np.testing.assert_allclose(d.numpy(), want_d.numpy(), rtol=0, atol=2e-3 * float(want_d.abs().max()))
    np.testing.assert_allclose(y_logits.numpy()[:, :, ::4, ::4], g[key + '__y_logits_sub4'], rtol=2e-4, atol=2e-5)
    pert, _ = ov.vat_perturbation(fnet, x, x_hat, eps0, vat_radius=case['vat_radius'], adaptive=case['adaptive'],
                                  cons_loss_fn=case['loss_fn'])
    want_p = torch.from_numpy(g[key + '__perturbation'])
    np.testing.assert_allclose(pert.reshape(2, -1).norm(dim=1).numpy(), case['pert_norm'], rtol=1e-5)      # the radius
    np.testing.assert_allclose(pert.numpy(), want_p.numpy(), rtol=0, atol=2e-3 * float(want_p.abs().max()))
