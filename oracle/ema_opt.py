"""
Oracle (test infrastructure): teacher EMA, Adam/SGD with duplicated parameter entries, LR schedules.

  ema_step()        -- EMAWeightOptimizer.step, optim_weight_ema.py:21-25. Three fp32 roundings
                       (t*alpha ; s*(1-alpha) ; sum), `1-alpha` formed in Python double then cast to fp32
                       by the tensor-scalar multiply. No FMA.
  adam_k_updates()  -- torch.optim.Adam's single-tensor update applied k times in a row with the same
                       gradient, which is what the reference gets for backbone weights because
                       `pretrained_parameters()` (deeplab2.py:208-230) yields them 3x / 4x
                       (SURVEY.md Appendix A, Q2). Defaults betas=(0.9, 0.999), eps=1e-8, no weight decay
                       (train_seg_semisup_mask_mt.py:90-93).
  sgd_k_updates()   -- torch.optim.SGD (momentum / nesterov / weight decay), same k-fold semantics
                       (train_seg_semisup_mask_mt.py:94-98).
  poly_lr(), cosine_lr(), multistep_lr() -- lr_schedules.py:24-35 and the torch schedulers it wires up
                       (lr_schedules.py:39-64), as closed forms of the explicit index the trainer passes
                       (train_seg_semisup_mask_mt.py:258-259, 288-289).
  sigmoid_rampup()  -- network_architectures.py:122-130.

Pinned by tests/golden/ema_*.npz, adam_*.npz, lr_*.npz.
"""
import math
import numpy as np


def ema_step(tgt, src, alpha):
    """tgt, src: float32 numpy arrays. Returns the new target (fp32, rounding order as the reference)."""
    a = np.float32(alpha)
    b = np.float32(1.0 - alpha)
    t = (tgt.astype(np.float32) * a).astype(np.float32)
    s = (src.astype(np.float32) * b).astype(np.float32)
    return (t + s).astype(np.float32)


def adam_k_updates(p, g, m, v, step, lr, k=1, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """
    fp32 numpy restatement of torch's `_single_tensor_adam` (no amsgrad, not maximize), applied k times.
    Rounding order follows torch: m.lerp_(g, 1-b1); v = v*b2 + (1-b2)*g*g (addcmul);
    denom = sqrt(v)/sqrt(bc2) + eps; p += -(lr/bc1) * m/denom.
    Returns (p, m, v, step).
    """
    p = p.astype(np.float32).copy()
    m = m.astype(np.float32).copy()
    v = v.astype(np.float32).copy()
    g = g.astype(np.float32)
    f32 = np.float32
    for _ in range(k):
        step += 1
        gg = g
        if weight_decay != 0.0:
            gg = (g + f32(weight_decay) * p).astype(np.float32)
        # exp_avg.lerp_(grad, 1 - beta1):  m + w*(g - m), w < 0.5 branch of lerp
        m = (m + f32(1.0 - beta1) * (gg - m)).astype(np.float32)
        v = (v * f32(beta2)).astype(np.float32)
        # ATen addcmul: self + (value * t1) * t2, evaluated left to right in fp32
        v = (v + ((f32(1.0 - beta2) * gg).astype(np.float32) * gg).astype(np.float32)).astype(np.float32)
        bc1 = 1.0 - beta1 ** step
        bc2 = 1.0 - beta2 ** step
        step_size = lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        denom = (np.sqrt(v) / f32(bc2_sqrt)).astype(np.float32) + f32(eps)
        # ATen addcdiv: self + (value * t1) / t2
        p = (p + ((f32(-step_size) * m).astype(np.float32) / denom).astype(np.float32)).astype(np.float32)
    return p, m, v, step


def sgd_k_updates(p, g, buf, lr, k=1, momentum=0.9, nesterov=False, weight_decay=0.0):
    """
    torch.optim.SGD single-tensor update (dampening 0) applied k times. Returns (p, buf).

    Quirk of torch >= 1.5/2.x with a duplicated entry (pinned by tests/golden/optim.npz, generated with the torch
    in this image): the momentum-buffer list is gathered BEFORE the loop, so on the very first step (`buf is
    None`) every one of the k visits sees "no buffer" and re-initialises it to its own d; from the second step on
    the k visits share one buffer and update it in place. (PyTorch 1.4 -- the reference's pin -- looked the state
    up per visit and also applied weight decay in place on .grad; that variant is not reproducible with the torch
    available here and is documented in DESIGN.md as unpinned.)
    """
    f32 = np.float32
    p = p.astype(np.float32).copy()
    g = g.astype(np.float32)
    first = buf is None
    buf = None if buf is None else buf.astype(np.float32).copy()
    for _ in range(k):
        d = g
        if weight_decay != 0.0:
            d = (g + f32(weight_decay) * p).astype(np.float32)
        if momentum != 0.0:
            if first:
                buf = d.copy()
            else:
                buf = ((buf * f32(momentum)).astype(np.float32) + d).astype(np.float32)
            d = (d + f32(momentum) * buf).astype(np.float32) if nesterov else buf
        p = (p + f32(-lr) * d).astype(np.float32)
    return p, buf


def poly_lr(base_lr, it, t_max, power=0.9, eta_min=0.0):
    if it == 0:
        return base_lr
    prog = min(max(float(it) / float(t_max), 0), 1)
    return base_lr * max((1.0 - prog) ** power, eta_min)


def cosine_lr(base_lr, it, t_max, eta_min=0.0):
    """Closed form that torch's CosineAnnealingLR uses when `.step(epoch)` is given an explicit index."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * it / t_max)) / 2


def multistep_lr(base_lr, epoch, milestones, gamma):
    n = sum(1 for ms in sorted(milestones) if ms <= epoch)
    return base_lr * gamma ** n


def sigmoid_rampup(current, rampup_length):
    if rampup_length == 0:
        return 1.0
    cur = float(np.clip(current, 0.0, rampup_length))
    ph = 1.0 - cur / rampup_length
    return float(np.exp(-5.0 * ph * ph))
