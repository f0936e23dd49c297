"""
Oracle (test infrastructure): CutMix paste, softmax / confidence, the five consistency losses, supervised CE.

PyTorch-CPU restatement (fp32 by default, fp64 on request) of the inline training-loop code of
train_seg_semisup_mask_mt.py; gradients come from torch autograd. `F.interpolate(..., align_corners=...)`,
`softmax`, `log_softmax` are the same ATen CPU ops the reference calls, so they are the reference itself run
here; everything else is explicit arithmetic.

  paste()              -- train_seg_semisup_mask_mt.py:350-351, 363   (x0*(1-m) + x1*m)
  upsample()           -- architectures/deeplab2.py:204  (bilinear, align_corners=True)
                          architectures/deeplab3plus.py:77 (align_corners=False)
  per_pixel_loss()     -- :428-446  (var / logits_var / logits_smoothl1 / bce / kld, summed over classes)
                          with robust_binary_crossentropy from architectures/network_architectures.py:115-118
  confidence()         -- :407-418
  consistency()        -- :407-458  (shared tail of mix and cut mode)
  mix_mode_loss()      -- :346-369 + consistency()
  cut_mode_loss()      -- :385-401 + consistency()
  supervised_ce()      -- :126, :299-300  (CrossEntropyLoss(ignore_index=255), mean over valid pixels)

Pinned by tests/golden/losses_*.npz.
"""
import math
import torch
import torch.nn.functional as F

LOSS_FNS = ('var', 'logits_var', 'logits_smoothl1', 'bce', 'kld')


def paste(x0, x1, m):
    return x0 * (1 - m) + x1 * m


def upsample(logits, size, align_corners=True):
    return F.interpolate(logits, size=tuple(size), mode='bilinear', align_corners=align_corners)


def per_pixel_loss(l_stu, l_tea, loss_fn):
    """(N,C,H,W) x2 -> (N,1,H,W), already summed over the class axis."""
    n_classes = l_stu.shape[1]
    root_c = math.sqrt(n_classes)
    if loss_fn == 'var':
        d = F.softmax(l_stu, dim=1) - F.softmax(l_tea, dim=1)
        out = (d * d).sum(dim=1, keepdim=True)
    elif loss_fn == 'logits_var':
        d = l_stu - l_tea
        out = (d * d).sum(dim=1, keepdim=True) / root_c
    elif loss_fn == 'logits_smoothl1':
        d = (l_stu - l_tea).abs()
        sl1 = torch.where(d < 1.0, 0.5 * d * d, d - 0.5)
        out = sl1.sum(dim=1, keepdim=True) / root_c
    elif loss_fn == 'bce':
        eps = 1e-6
        p = F.softmax(l_stu, dim=1)
        t = F.softmax(l_tea, dim=1)
        out = (-(t * torch.log(p + eps) + (1.0 - t) * torch.log(1.0 - p + eps))).sum(dim=1, keepdim=True)
    elif loss_fn == 'kld':
        t = F.softmax(l_tea, dim=1)
        logp = F.log_softmax(l_stu, dim=1)
        # F.kl_div pointwise = xlogy(t, t) - t * logp   (0 where t == 0)
        out = (torch.xlogy(t, t) - t * logp).sum(dim=1, keepdim=True)
    else:
        raise ValueError('Unknown consistency loss function {}'.format(loss_fn))
    return out


def confidence(l_tea, conf_thresh):
    """-> (conf_mask (N,1,H,W) float, rate scalar tensor)."""
    conf = F.softmax(l_tea, dim=1).max(dim=1)[0]
    cm = (conf >= conf_thresh).to(l_tea.dtype)[:, None, :, :]
    return cm, cm.mean()


def consistency(l_stu, l_tea, loss_mask, loss_fn='var', conf_thresh=0.97, conf_per_pixel=False,
                ramp_val=1.0, rampup=-1, cons_weight=1.0):
    """
    Returns dict(consistency_loss=<value the reference logs, :461>, unsup_loss=<value it back-props, :458>,
                 conf_rate=<:413 or None>).
    """
    rate = None
    if conf_thresh > 0.0:
        cm, rate = confidence(l_tea.detach(), conf_thresh)
        loss_mask = loss_mask * (cm if conf_per_pixel else rate)
    pix = per_pixel_loss(l_stu, l_tea.detach(), loss_fn)
    closs = (pix * loss_mask).mean()
    if rampup > 0:
        closs = closs * ramp_val
    return dict(consistency_loss=closs, unsup_loss=closs * cons_weight, conf_rate=rate)


def mix_mode_loss(l_stu, l0_tea, l1_tea, m, um0, um1, **kw):
    """Mix mode: teacher logits and validity masks are pasted with the same box mask as the images."""
    l_tea = paste(l0_tea, l1_tea, m)
    um = paste(um0, um1, m)
    return consistency(l_stu, l_tea, um, **kw)


def cut_mode_loss(l_stu, l_tea, m, um, **kw):
    """Cut ('zero') mode: loss only where the image was kept (m == 1) and valid."""
    return consistency(l_stu, l_tea, m * um, **kw)


def supervised_ce(logits, labels, ignore_index=255):
    """logits (N,C,H,W); labels int64 (N,H,W). Mean of -log_softmax[label] over labels != ignore_index."""
    logp = F.log_softmax(logits, dim=1)
    valid = labels != ignore_index
    safe = torch.where(valid, labels, torch.zeros_like(labels))
    picked = torch.gather(logp, 1, safe[:, None]).squeeze(1)
    n_valid = valid.sum()
    return -(picked * valid.to(logp.dtype)).sum() / n_valid.to(logp.dtype)
