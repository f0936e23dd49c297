"""
The explicit forward / backward chain of oracle/deeplab2_chain.py (the restatement the bf16 configuration of the device
engine is held to) against the PINNED fp32 oracle: with the storage model switched off the chain must reproduce
oracle/deeplab2.py + torch autograd (which tests/test_oracle_golden.py ties to the reference-generated fixtures), tensor by
tensor; the whole iteration through the chain must reproduce oracle/step.py (pinned by tests/golden/step.npz).
"""
import numpy as np
import pytest
import torch

from oracle import deeplab2 as dl, deeplab2_chain as ch, step as ostep, boxmask as obox


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize('layers,shape', [([1, 1, 1, 1], (2, 3, 33, 33)), ([2, 1, 2, 1], (1, 3, 41, 57))])
def test_chain_without_storage_model_equals_autograd_of_the_pinned_oracle(layers, shape):
    C = 5
    st = dl.closed_form_state(C, layers)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(*shape, generator=g)
    keys = dl.trainable_keys(C, layers)
    leaves = {k: st[k].clone().requires_grad_(True) for k in keys}
    s2 = dict(st)
    s2.update(leaves)
    lo = dl.forward_lowres(x, s2, layers)
    dlg = torch.randn(lo.shape, generator=g)
    lo.backward(dlg)
    c = ch.Chain(st, C, layers, 'fp32')
    lo2, saved = c.forward(x)
    assert _rel(lo2, lo.detach()) <= 5e-6
    grads = c.backward(saved, dlg)
    for k in keys:
        want = leaves[k].grad
        if want is None:                       # ASPP d18 / d24
            assert k not in grads
            continue
        assert _rel(grads[k], want) <= 2e-5, k
    # the saved block inputs are the oracle's own taps
    taps = {}
    dl.forward_lowres(x, st, layers, taps=taps)
    assert _rel(c.block_outputs(saved)[-1], taps['layer4']) <= 5e-6
    assert _rel(c.block_outputs(saved)[0], taps['stem']) <= 5e-6


def _iteration(storage, seed=5):
    C, layers, N, H, W = 5, [1, 1, 1, 1], 2, 33, 33
    st = dl.closed_form_state(C, layers)
    g = torch.Generator().manual_seed(seed)
    rnd = lambda: torch.randn(N, 3, H, W, generator=g).bfloat16().float()
    x, ux0, ux1 = rnd(), rnd(), rnd()
    y = torch.randint(0, C, (N, 1, H, W), generator=g)
    y[torch.rand(N, 1, H, W, generator=g) < 0.05] = 255
    ranges = obox.generate_ranges(np.random.RandomState(1), N, (H, W), (0.5, 0.5), 1) if hasattr(obox, 'generate_ranges') else None
    if ranges is None:
        m = (torch.rand(N, 1, H, W, generator=g) < 0.5).float()
    else:
        m = torch.tensor(obox.rasterise(ranges, (H, W), True).astype(np.float32))
    ones = torch.ones(N, 1, H, W)
    S = ostep.StepState(st, C, layers, opt='adam', lr=1e-3, teacher_alpha=0.99)
    grads = {}
    r = ostep.train_iteration(S, x, y, ux0, ux1, ones, ones, m, conf_thresh=0.3, grads_out=grads, storage=storage)
    return r, grads, S


def test_iteration_through_the_chain_equals_the_pinned_step():
    r0, g0, S0 = _iteration(None)
    r1, g1, S1 = _iteration('fp32')
    for k in ('sup_loss', 'consistency_loss', 'conf_rate'):
        assert r1[k] == pytest.approx(r0[k], rel=2e-5, abs=1e-7), k
    for k, want in g0.items():
        if want is None:
            assert g1[k] is None
        else:
            assert _rel(g1[k], want) <= 5e-5, k
    # the update itself is Adam's sign-like first step: compare the moments (linear in the gradient), and the weights loosely
    for k in S0.m:
        assert _rel(S1.m[k], S0.m[k]) <= 5e-5 or float(S0.m[k].abs().max()) == 0.0, k
        assert torch.allclose(S1.student[k], S0.student[k], rtol=0, atol=2.5e-4), k


def test_bf16_storage_model_moves_the_iteration_by_storage_noise_only():
    r0, g0, _ = _iteration('fp32')
    r1, g1, _ = _iteration('bf16')
    assert r1['sup_loss'] == pytest.approx(r0['sup_loss'], rel=2e-2)
    assert abs(r1['conf_rate'] - r0['conf_rate']) <= 2e-2
    rels = [_rel(g1[k], g0[k]) for k in g0 if g0[k] is not None]
    assert 1e-4 < max(rels) < 0.2 and float(np.mean(rels)) < 5e-2      # it rounds, and only rounds
    # rounded values are bf16-representable where the model says they are stored
    C, layers = 5, [1, 1, 1, 1]
    st = dl.closed_form_state(C, layers)
    c = ch.Chain(st, C, layers, 'bf16')
    _, saved = c.forward(torch.randn(1, 3, 33, 33).bfloat16().float())
    for t in c.block_outputs(saved):
        assert torch.equal(t, t.bfloat16().float())


# ---------------------------------------------------------------------------------------------------------- DeepLab v3+ units
def test_v3plus_unit_chain_equals_the_v3plus_oracle_with_storage_off():
    """oracle/deeplab3plus_chain.py assembles the DeepLab v3+ forward from its units (raw convolution; BatchNorm + residual +
    ReLU): with the storage model switched off it must BE oracle/deeplab3plus.py -- frozen and batch-statistics BatchNorm."""
    from oracle import deeplab3plus as o3, deeplab3plus_chain as oc
    layers, C = (1, 1, 2, 1), 5
    # seeded He weights: the closed-form fixture weights leave some channels almost constant over a batch, where fp32
    # batch_norm's own variance is rounding noise (the two statements then differ by that noise, 3e-4)
    gs = torch.Generator().manual_seed(4321)
    st = {}
    for k, (shape, dt) in o3.state_spec(C, layers).items():
        if dt == torch.int64:
            st[k] = torch.zeros(shape, dtype=torch.int64)
        elif len(shape) == 4:
            st[k] = torch.randn(shape, generator=gs) * (2.0 / (shape[1] * shape[2] * shape[3])) ** 0.5
        elif k.endswith('running_var'):
            st[k] = 0.8 + 0.4 * torch.rand(shape, generator=gs)
        elif k.endswith('running_mean'):
            st[k] = 0.1 * torch.randn(shape, generator=gs)
        elif k.endswith('.weight'):
            st[k] = 0.6 + 0.8 * torch.rand(shape, generator=gs)
        else:
            st[k] = 0.1 * torch.randn(shape, generator=gs)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(4, 3, 65, 81, generator=g)
    for bf, hf in ((True, True), (True, False), (False, False)):
        want = o3.forward_lowres(x, st, layers, backbone_frozen=bf, head_frozen=hf)
        taps = []
        got = oc.forward_lowres(x, st, layers, backbone_frozen=bf, head_frozen=hf, storage='fp32', taps=taps)
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-6, (bf, hf, float((got - want).abs().max()))
        assert len(taps) == 1 + sum(layers) * 3 + 4 + 9              # stem, 3 units per bottleneck, 4 shortcuts, 9 head units


@pytest.mark.parametrize('frozen', [False, True], ids=['batch_statistics', 'frozen'])
@pytest.mark.parametrize('groups', [1, 2])
def test_v3plus_unit_backward_equals_autograd(frozen, groups):
    """The written-out backward of the two unit kinds against ATen's autograd of the same unit (fp32 storage): data gradient
    and weight gradient of the convolution; du / dres / dgamma / dbeta of BatchNorm (+ residual) + ReLU, also with two sample
    groups normalised apart."""
    from oracle import deeplab3plus_chain as oc
    if frozen and groups > 1:
        pytest.skip('sample groups only matter for batch statistics')
    g = torch.Generator().manual_seed(8)
    x = torch.randn(4, 6, 9, 11, generator=g, requires_grad=True)
    w = (torch.randn(8, 6, 3, 3, generator=g) * 0.2).requires_grad_(True)
    du = torch.randn(4, 8, 5, 6, generator=g)
    u = torch.nn.functional.conv2d(x, w, None, 2, 2, 2)
    assert u.shape == du.shape
    gx, gw = torch.autograd.grad(u, (x, w), du)
    dx, dw = oc.conv_unit_backward(x.detach(), w.detach(), du, 2, 2, 2, storage='fp32')
    torch.testing.assert_close(dx, gx, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dw, gw, rtol=1e-5, atol=1e-5)
    # BatchNorm + residual + ReLU
    uu = torch.randn(4, 8, 5, 6, generator=g, requires_grad=True)
    res = torch.randn(4, 8, 5, 6, generator=g, requires_grad=True)
    gamma = (torch.rand(8, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(8, generator=g) * 0.2).requires_grad_(True)
    rm, rv = torch.randn(8, generator=g) * 0.1, torch.rand(8, generator=g) + 0.5
    dy = torch.randn(4, 8, 5, 6, generator=g)
    parts = []
    for k in range(groups):
        sl = slice(k * 4 // groups, (k + 1) * 4 // groups)
        parts.append(torch.nn.functional.batch_norm(uu[sl], rm.clone(), rv.clone(), gamma, beta, not frozen, 0.1, 1e-5))
    yref = torch.relu(torch.cat(parts, 0) + res)
    gu, gres, gg, gb = torch.autograd.grad(yref, (uu, res, gamma, beta), dy)
    y, ctx = oc.bn_unit(uu.detach(), gamma.detach(), beta.detach(), rm, rv, True, res.detach(), frozen, 'fp32', groups)
    torch.testing.assert_close(y, yref.detach(), rtol=1e-5, atol=1e-5)
    d_u, d_res, d_g, d_b = oc.bn_unit_backward(uu.detach(), y, dy, gamma.detach(), ctx, True, True, frozen, 'fp32')
    torch.testing.assert_close(d_u, gu, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(d_res, gres, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(d_g, gg, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(d_b, gb, rtol=1e-4, atol=1e-4)
    if not frozen and groups == 1:                        # the running statistics move like nn.BatchNorm2d's
        rm2, rv2 = rm.clone(), rv.clone()
        torch.nn.functional.batch_norm(uu.detach(), rm2, rv2, gamma.detach(), beta.detach(), True, 0.1, 1e-5)
        torch.testing.assert_close(ctx['running_mean'], rm2, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(ctx['running_var'], rv2, rtol=1e-5, atol=1e-6)
