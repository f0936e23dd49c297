"""
Virtual adversarial training (VAT) mean-teacher iteration -- the body of the reference's second trainer,
train_seg_semisup_vat_mt.py:213-301 (direction / perturbation) and :346-476 (iteration); SURVEY.md 8(f) rank 2.

    student(x_sup) -> CE -> backward                                                    :356-358
    x_perturb = vat_perburbation(x_tea, x_stu)                                          :398   (see below)
    teacher(x_tea) (no grad), student(x_stu + x_perturb)                                :404-407
    consistency between the two, masked by the validity mask x confidence               :413-459
    optimizer + EMA                                                                     :476-478

What runs where:
  * the two network passes of the ITERATION, the cross entropy, the consistency loss (the fused kernel in "cut" mode with
    an all-ones box mask: loss_mask = um * confidence, exactly :415-429) and Adam/SGD + EMA are the same MI355X kernels
    as in the CutMix step (step.py);
  * the VAT DIRECTION is one step of power iteration: the gradient, wrt a perturbation of norm 1e-6*H*W/1000, of the
    distance between net(x_hat + eps) and net(x) (:244-268). That perturbation is ~2e-7 per pixel -- below bf16
    resolution of the input (and barely above fp32's); the pass therefore runs the network in fp32, whatever
    `compute_dtype` the iteration uses: for the DeepLab networks on the fp32 configuration of the hand-written engine
    (f32-input MFMA convolutions csrc/conv_f32.hip, stem incl. its image gradient csrc/stem.hip -- no library
    convolution in the pass), for the U-Nets on the library's fp32 convolutions with csrc/bn.hip BatchNorm. The
    distance and its gradient wrt the low-resolution logits come
    from the fused consistency kernels (the reference SUMS where the kernel averages; the direction is normalised, so a
    positive factor is immaterial), the gradient wrt the image from the network's backward pass with the weight
    gradients switched off (`torch.autograd.grad` wrt eps only, like the reference);
  * quirk kept: the network used for the direction is put in eval mode and stays there (:237).
"""
import math

import torch

from . import ops


def normalize_eps(x):
    """x / (per-sample L2 norm + 1e-12), train_seg_semisup_vat_mt.py:216-219."""
    flat = x.reshape(len(x), -1)
    mag = torch.sqrt((flat * flat).sum(dim=1))
    return x / (mag[:, None, None, None] + 1e-12)


def normalized_noise_like(x, scale=1.0, generator=None):
    """:221-226 (without the requires_grad switch: the direction pass sets it)."""
    eps = torch.randn(x.shape, dtype=torch.float32, device=x.device, generator=generator)
    return normalize_eps(eps) * scale


def _ones_ranges(n, device):
    # an empty box with invert=False rasterises to an all-ones mask (mask_gen.py:110-116)
    return torch.zeros((n, 1, 4), dtype=torch.int32, device=device)


class _DataGradOnly(object):
    """Backward passes inside this context produce input gradients only (no weight gradients into the arenas)."""

    def __init__(self, net):
        self.net = net

    def __enter__(self):
        # read by whichever executor of the network runs the backward pass (backbone_hip.py)
        self.prev = getattr(self.net, '_data_grad_only', False)
        self.net._data_grad_only = True

    def __exit__(self, *exc):
        self.net._data_grad_only = self.prev


def vat_direction(net, x, x_hat, cons_loss_fn='kld', eps0=None, generator=None):
    """
    -> (normalised adversarial direction fp32 (N,C,H,W), low-resolution logits of net(x) in eval mode).
    `eps0` (optional): the initial noise, already normalised and scaled (tests pass the oracle's draw).
    """
    if cons_loss_fn not in ('var', 'bce', 'kld', 'logits_var'):
        raise ValueError('Unknown consistency loss function {}'.format(cons_loss_fn))
    net.eval()                                                  # and it stays there (:237)
    out_size = x.shape[2:4]
    align = getattr(net, 'upsample_align_corners', True)
    prev_dtype = getattr(net, 'compute_dtype', None)
    try:
        if prev_dtype is not None:
            net.compute_dtype = torch.float32                  # the perturbation is below bf16 resolution
        with torch.no_grad():
            y_lo = net.forward_lowres(x.float())
        if eps0 is None:
            eps0 = normalized_noise_like(x, 1.0e-6 * x.shape[2] * x.shape[3] / 1000, generator)
        eps = eps0.detach().clone().float().requires_grad_(True)
        with torch.enable_grad(), _DataGradOnly(net):
            e_lo = net.forward_lowres(x_hat.detach().float() + eps)
            cfg = ops.ConsistencyConfig(mode='cut', loss_fn=cons_loss_fn, conf_thresh=0.0, conf_per_pixel=False,
                                        align_corners=align, invert=False)
            sc, ctx = ops.consistency_forward(cfg, e_lo.detach(), y_lo, None, out_size,
                                              ranges=_ones_ranges(x.shape[0], x.device), sync_conf_rate=False)
            g_lo = ops.consistency_backward(ctx, sc)
            g, = torch.autograd.grad(outputs=e_lo, inputs=eps, grad_outputs=g_lo.to(e_lo.dtype))
    finally:
        if prev_dtype is not None:
            net.compute_dtype = prev_dtype
    return normalize_eps(g.float()), y_lo


def vat_radius_of(x_hat, vat_radius, adaptive):
    """:277-299 -- a scalar, or a per-sample (N,1,1,1) tensor from the image Jacobian."""
    if adaptive:
        xf = x_hat.float()
        dv = (xf[:, :, 2:, :] - xf[:, :, :-2, :]).reshape(len(xf), -1)
        dh = (xf[:, :, :, 2:] - xf[:, :, :, :-2]).reshape(len(xf), -1)
        return vat_radius * torch.sqrt((dv ** 2).sum(dim=1) + (dh ** 2).sum(dim=1))[:, None, None, None] * 0.5
    return vat_radius * math.sqrt(float(x_hat.shape[1] * x_hat.shape[2] * x_hat.shape[3]))


def vat_perturbation(net, x, x_hat, vat_radius=0.5, adaptive=False, cons_loss_fn='kld', eps0=None, generator=None):
    d, y_lo = vat_direction(net, x, x_hat, cons_loss_fn, eps0, generator)
    return (d * vat_radius_of(x_hat, vat_radius, adaptive)).detach(), y_lo


class VATConfig(object):
    def __init__(self, vat_radius=0.5, adaptive_vat_radius=False, cons_loss_fn='kld', cons_weight=1.0, conf_thresh=0.97,
                 conf_per_pixel=False, rampup=-1, unsup_batch_ratio=1):
        if cons_loss_fn not in ('var', 'bce', 'kld', 'logits_var'):
            raise ValueError('Unknown consistency loss function {}'.format(cons_loss_fn))
        self.vat_radius = float(vat_radius)
        self.adaptive = bool(adaptive_vat_radius)
        self.cons_loss_fn = cons_loss_fn
        self.cons_weight = float(cons_weight)
        self.rampup = rampup
        self.unsup_batch_ratio = int(unsup_batch_ratio)
        self.cons = ops.ConsistencyConfig(mode='cut', loss_fn=cons_loss_fn, conf_thresh=conf_thresh,
                                          conf_per_pixel=conf_per_pixel, invert=False)


class VATUnsupBatch(object):
    """x_tea / x_stu: teacher (weakly) and student (strongly) augmented images; um: validity mask (N,1,H,W) or None."""

    def __init__(self, x_tea, x_stu=None, um=None):
        self.x_tea = x_tea
        self.x_stu = x_tea if x_stu is None else x_stu
        self.um = um


class VATMeanTeacherStep(object):
    def __init__(self, student_net, teacher_net, student_optim, teacher_optim, cfg, vat_dir_from_student=False,
                 group=None, generator=None):
        self.student, self.teacher = student_net, teacher_net
        self.student_optim, self.teacher_optim = student_optim, teacher_optim
        self.cfg = cfg
        self.vat_dir_net = student_net if vat_dir_from_student else teacher_net       # :102-105
        self.group = group
        self.generator = generator
        self.align_corners = getattr(student_net, 'upsample_align_corners', True)
        cfg.cons.align_corners = self.align_corners
        import torch.distributed as dist
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1

    def __call__(self, sup_x, sup_y, unsup_batches, ramp_val=1.0, eps0=None):
        cfg = self.cfg
        out_size = sup_x.shape[2:4]
        self.student_optim.zero_grad()
        ramp = ramp_val if cfg.rampup > 0 else 1.0
        lo = self.student.forward_lowres(sup_x)
        ce_sc, ce_ctx = ops.ce_forward(lo.detach(), sup_y, out_size, 255, self.align_corners, group=self.group)
        lo.backward(ops.ce_backward(ce_ctx, ce_sc).to(lo.dtype))
        cons_vals = []
        if cfg.cons_weight > 0.0:
            for ub in unsup_batches:
                x_perturb, _ = vat_perturbation(self.vat_dir_net, ub.x_tea, ub.x_stu, cfg.vat_radius, cfg.adaptive,
                                                cfg.cons_loss_fn, eps0=eps0, generator=self.generator)
                x_adv = (ub.x_stu.float() + x_perturb).to(ub.x_stu.dtype)
                with torch.no_grad():
                    l_tea = self.teacher.forward_lowres(ub.x_tea)
                l_stu = self.student.forward_lowres(x_adv)
                sc, cctx = ops.consistency_forward(cfg.cons, l_stu.detach(), l_tea, None, out_size,
                                                   ranges=_ones_ranges(x_adv.shape[0], x_adv.device), um0=ub.um,
                                                   ramp_val=ramp, cons_weight=cfg.cons_weight, group=self.group)
                l_stu.backward(ops.consistency_backward(cctx, sc).to(l_stu.dtype))
                cons_vals.append(sc)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.student_optim.arena.grad, op=dist.ReduceOp.SUM, group=self.group)
            self.student_optim.grad_scale = 1.0 / self.world
        self.student_optim.step()
        if self.teacher_optim is not None:
            self.teacher_optim.step()
        res = dict(sup_loss=ce_sc[0], consistency_loss=None, conf_rate=None)
        if cons_vals:
            stacked = torch.stack(cons_vals)
            res['consistency_loss'] = stacked[:, 0].mean()
            res['conf_rate'] = stacked[:, 1].mean()
        return res
