"""
Oracle (test infrastructure): the virtual-adversarial pieces of train_seg_semisup_vat_mt.py on PyTorch-CPU fp32
(SURVEY.md 8(f) rank 2).

PINNED (round 4): in the reference these are closures inside the 570-line trainer function (`t_dot`, `normalize_eps`,
`normalized_noise_like`, `vat_direction`, `vat_perburbation`, train_seg_semisup_vat_mt.py:213-301), which cannot be
imported. tests/golden/make_golden.py::gen_vat cuts their FunctionDef nodes out of the reference's source with `ast` at
generation time, binds the trainer's free variables to a tiny reference DeepLab v2 and runs them; tests/golden/vat.npz holds
the outputs (initial noise as drawn, direction, perturbation, teacher logits; 4 consistency functions x fixed / adaptive
radius) and tests/test_oracle_golden.py::test_vat_direction_and_perturbation_vs_the_reference_closures holds this file to
them. What the functions do:

  normalize_eps(x)          x / (||x||_2 per sample + 1e-12)                                            :216-219
  vat_direction(x, x_hat)   y = net_eval(x) (no grad); eps0 = normalised noise * 1e-6*H*W/1000;          :227-271
                            loss = SUM over everything of the consistency function between net_eval(x_hat + eps0) and y;
                            direction = normalize_eps(d loss / d eps0)
  vat_perturbation          direction * radius; radius = vat_radius * sqrt(C*H*W), or adaptive:          :274-301
                            vat_radius * 0.5 * sqrt(sum (x_hat[2:] - x_hat[:-2])^2 over both image axes) per sample
  the network used for the direction is put in eval mode and never put back within the epoch             :237

The network is passed as a callable `net(x) -> logits (N,C,H,W)` (e.g. a closure over oracle.deeplab2.forward).
"""
import math

import torch
import torch.nn.functional as F


def normalize_eps(x):
    flat = x.reshape(len(x), -1)
    mag = torch.sqrt((flat * flat).sum(dim=1))
    return x / (mag[:, None, None, None] + 1e-12)


def noise_scale(shape):
    return 1.0e-6 * shape[2] * shape[3] / 1000


def direction_loss(eps_logits, y_logits, cons_loss_fn):
    """The SUM-reduced consistency function of vat_direction (:251-263)."""
    y_prob = F.softmax(y_logits, dim=1)
    if cons_loss_fn == 'var':
        d = F.softmax(eps_logits, dim=1) - y_prob
        return (d * d).sum()
    if cons_loss_fn == 'bce':
        p = F.softmax(eps_logits, dim=1)
        return (-(y_prob * torch.log(p + 1e-6) + (1.0 - y_prob) * torch.log(1.0 - p + 1e-6))).sum()
    if cons_loss_fn == 'kld':
        return F.kl_div(F.log_softmax(eps_logits, dim=1), y_prob, reduction='none').sum()
    if cons_loss_fn == 'logits_var':
        d = eps_logits - y_logits
        return (d * d).sum()
    raise ValueError('Unknown consistency loss function {}'.format(cons_loss_fn))


def vat_direction(net, x, x_hat, eps0, cons_loss_fn='kld'):
    """`eps0`: the normalised, scaled initial noise (the reference draws it with torch.randn; passed in here so that
    device and oracle use the same draw). Returns (direction, y_logits)."""
    with torch.no_grad():
        y_logits = net(x)
    eps = eps0.clone().detach().requires_grad_(True)
    loss = direction_loss(net(x_hat.detach() + eps), y_logits, cons_loss_fn)
    g, = torch.autograd.grad(loss, eps)
    return normalize_eps(g), y_logits


def vat_radius_of(x_hat, vat_radius, adaptive):
    if adaptive:
        dv = (x_hat[:, :, 2:, :] - x_hat[:, :, :-2, :]).reshape(len(x_hat), -1)
        dh = (x_hat[:, :, :, 2:] - x_hat[:, :, :, :-2]).reshape(len(x_hat), -1)
        return vat_radius * torch.sqrt((dv ** 2).sum(dim=1) + (dh ** 2).sum(dim=1))[:, None, None, None] * 0.5
    return vat_radius * math.sqrt(float(x_hat.shape[1] * x_hat.shape[2] * x_hat.shape[3]))


def vat_perturbation(net, x, x_hat, eps0, vat_radius=0.5, adaptive=False, cons_loss_fn='kld'):
    d, y_logits = vat_direction(net, x, x_hat, eps0, cons_loss_fn)
    return (d * vat_radius_of(x_hat, vat_radius, adaptive)).detach(), y_logits
