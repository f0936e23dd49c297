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
