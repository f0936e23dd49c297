"""
Oracle (test infrastructure): ONE whole CutMix mean-teacher training iteration on PyTorch-CPU fp32, restated from
train_seg_semisup_mask_mt.py:287-467 with the oracle pieces:

    zero grads                                             :290
    sup:   CE(student(x), y[:,0]).backward()               :296-301
    unsup: paste images / validity masks                   :346-351   (mix)   |  x*m   :389 (cut)
           teacher(x0), teacher(x1) under no_grad          :354-356
           student(x_mix)                                  :358
           paste teacher logits, softmax, confidence, loss :363-458
           backward                                        :459
    optimizer step (k-fold updates on duplicated entries)  :465   (deeplab2.py:208-230 duplicates)
    EMA step over every float state tensor                 :466-467 (optim_weight_ema.py:21-25)

This is what `bench.py` times as `cpu_baseline` (kind "port") and what tests compare the device step against.
State lives in plain dicts of tensors keyed by the reference's state_dict names.

Pinned by tests/golden/step.npz (three iterations of the reference modules on the tiny network).
"""
import math
from collections import OrderedDict

import torch

from . import deeplab2 as dl
from . import losses as L


class StepState(object):
    """Student / teacher state dicts + optimizer slots for the trainable tensors."""

    def __init__(self, student_state, num_classes, layers=dl.LAYERS, opt='adam', lr=1e-4, teacher_alpha=0.99,
                 sgd_momentum=0.9, sgd_nesterov=False, sgd_weight_decay=5e-4):
        self.layers = tuple(layers)
        self.num_classes = num_classes
        self.student = OrderedDict((k, v.clone()) for k, v in student_state.items())
        # EMAWeightOptimizer.__init__ copies every float tensor (optim_weight_ema.py:12-13); the int64
        # num_batches_tracked entries keep the teacher's own values.
        self.teacher = OrderedDict((k, v.clone()) for k, v in student_state.items())
        self.opt = opt
        self.teacher_alpha = teacher_alpha
        self.sgd = (sgd_momentum, sgd_nesterov, sgd_weight_decay)
        g0, g1 = dl.param_multiplicity(num_classes, layers)
        self.entries = [(k, mult, lr * 0.1) for k, mult in g0.items()] + [(k, mult, lr) for k, mult in g1.items()]
        self.base_lrs = {k: l for k, _, l in self.entries}
        self.lr_scale = 1.0
        self.m = {k: torch.zeros_like(self.student[k]) for k, _, _ in self.entries}
        self.v = {k: torch.zeros_like(self.student[k]) for k, _, _ in self.entries}
        self.steps = {k: 0 for k, _, _ in self.entries}
        self.buf = {k: None for k, _, _ in self.entries}


def _adam_k(p, g, m, v, step, lr, k, beta1=0.9, beta2=0.999, eps=1e-8):
    for _ in range(k):
        step += 1
        m.lerp_(g, 1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        bc1 = 1 - beta1 ** step
        bc2 = 1 - beta2 ** step
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))
    return step


def _sgd_k(p, g, buf, lr, k, momentum, nesterov, wd):
    first = buf is None       # see oracle/ema_opt.py:sgd_k_updates for the first-step quirk
    for _ in range(k):
        d = g
        if wd != 0:
            d = d.add(p, alpha=wd)
        if momentum != 0:
            if first:
                buf = d.clone()
            else:
                buf.mul_(momentum).add_(d)
            d = d.add(buf, alpha=momentum) if nesterov else buf
        p.add_(d, alpha=-lr)
    return buf


def train_iteration(S, sup_x, sup_y, ux0, ux1, um0, um1, masks, mode='mix', loss_fn='var', conf_thresh=0.97,
                    conf_per_pixel=False, ramp_val=1.0, rampup=-1, cons_weight=1.0, frozen_bn=True, grads_out=None,
                    pi_model=False, storage=None, taps_out=None):
    """
    One iteration. `sup_y` int64 (N,1,H,W) with 255 = ignore; `masks` float (N,1,H,W) in {0,1}.
    In cut mode ux1/um1 are ignored. Returns dict(sup_loss, consistency_loss, conf_rate).
    `grads_out` (dict, optional) receives the autograd gradient of every trainable tensor (what the optimizer consumed)
    -- the yardstick the tests hold the device backward pass to.
    `pi_model`: `--model pi` (train_seg_semisup_mask_mt.py:110-113): the teacher IS the student network (its forward
    passes run under no_grad with the student's current weights) and there is no EMA step.
    `storage`: None = the fp32 restatement through torch autograd (pinned by tests/golden/step.npz). 'bf16' / 'fp32' =
    the same iteration through the explicit forward / backward chain of oracle/deeplab2_chain.py, with ('bf16') or
    without ('fp32') the storage roundings of the device's throughput configuration -- losses, optimizer and EMA are the
    code below either way; `taps_out` (dict) then receives the block inputs of the student passes.
    """
    if storage is not None:
        return _train_iteration_chain(S, sup_x, sup_y, ux0, ux1, um0, um1, masks, mode, loss_fn, conf_thresh,
                                      conf_per_pixel, ramp_val, rampup, cons_weight, frozen_bn, grads_out, pi_model,
                                      storage, taps_out)
    keys = [k for k, _, _ in S.entries]
    leaves = {k: S.student[k].clone().requires_grad_(True) for k in keys}
    st = OrderedDict(S.student)
    st.update(leaves)
    frozen = bool(frozen_bn)

    def run(x, state, target):
        """One forward pass. Without --freeze_bn (the reference CLI's default, train_seg_semisup_mask_mt.py:587) every
        BatchNorm normalises with BATCH statistics and moves its running statistics (momentum 0.1) -- in the teacher too,
        which the loop keeps in train mode (:269-270); the affine parameters never train (deeplab2.py:72-84).
        The passes update `target`'s running statistics one after the other, in the reference's order."""
        if frozen:
            return dl.forward(x, state, S.layers, frozen=True)
        ns = {}
        out = dl.forward(x, state, S.layers, frozen=False, new_stats=ns)
        for k, v in ns.items():
            target[k] = v.detach()
            if state is not target:
                state[k] = target[k]
        return out

    logits_sup = run(sup_x, st, S.student)
    sup_loss = L.supervised_ce(logits_sup, sup_y[:, 0])
    total = sup_loss
    closs = None
    rate = None
    if cons_weight > 0.0:
        tea = st if pi_model else S.teacher
        tea_target = S.student if pi_model else S.teacher
        with torch.no_grad():
            l0 = run(ux0, tea, tea_target)
            l1 = run(ux1, tea, tea_target) if mode == 'mix' else None
        kw = dict(loss_fn=loss_fn, conf_thresh=conf_thresh, conf_per_pixel=conf_per_pixel, ramp_val=ramp_val,
                  rampup=rampup, cons_weight=cons_weight)
        if mode == 'mix':
            l_stu = run(L.paste(ux0, ux1, masks), st, S.student)
            r = L.mix_mode_loss(l_stu, l0, l1, masks, um0, um1, **kw)
        else:
            l_stu = run(ux0 * masks, st, S.student)
            r = L.cut_mode_loss(l_stu, l0, masks, um0, **kw)
        total = total + r['unsup_loss']
        closs, rate = r['consistency_loss'], r['conf_rate']
    total.backward()       # the reference back-props the two losses separately; the gradients add
    if grads_out is not None:
        for k in keys:
            grads_out[k] = None if leaves[k].grad is None else leaves[k].grad.detach().clone()

    _apply_updates(S, {k: leaves[k].grad for k in keys}, pi_model)
    return dict(sup_loss=float(sup_loss.detach()),
                consistency_loss=None if closs is None else float(closs.detach()),
                conf_rate=None if rate is None else float(rate))


def _apply_updates(S, grads, pi_model):
    """optimizer step with the duplicated entries (:465; deeplab2.py:208-230) + EMA (:466-467)."""
    with torch.no_grad():
        for k, mult, base_lr in S.entries:
            g = grads.get(k)
            if g is None:      # ASPP d18/d24 never receive gradients and are skipped by the optimizer
                continue
            lr = base_lr * S.lr_scale
            if S.opt == 'adam':
                S.steps[k] = _adam_k(S.student[k], g, S.m[k], S.v[k], S.steps[k], lr, mult)
            else:
                mom, nest, wd = S.sgd
                S.buf[k] = _sgd_k(S.student[k], g, S.buf[k], lr, mult, mom, nest, wd)
        a = S.teacher_alpha
        for k, t in ([] if pi_model else S.teacher.items()):
            if t.dtype == torch.float32:
                t.mul_(a)
                t.add_(S.student[k] * (1.0 - a))


def _train_iteration_chain(S, sup_x, sup_y, ux0, ux1, um0, um1, masks, mode, loss_fn, conf_thresh, conf_per_pixel,
                           ramp_val, rampup, cons_weight, frozen_bn, grads_out, pi_model, storage, taps_out=None):
    from . import deeplab2_chain as ch
    if not frozen_bn:
        raise NotImplementedError('oracle step covers the --freeze_bn configuration (cfg 2/3)')
    keys = [k for k, _, _ in S.entries]
    stu = ch.Chain(S.student, S.num_classes, S.layers, storage)
    size = tuple(sup_x.shape[2:4])

    def leaf(lo):
        return lo.detach().requires_grad_(True)

    lo_sup, sv_sup = stu.forward(sup_x)
    lo_sup = leaf(lo_sup)
    sup_loss = L.supervised_ce(L.upsample(lo_sup, size), sup_y[:, 0])
    total = sup_loss
    closs = rate = None
    lo_mix = sv_mix = None
    if cons_weight > 0.0:
        tea = stu if pi_model else ch.Chain(S.teacher, S.num_classes, S.layers, storage)
        l0 = L.upsample(tea.forward(ux0, save=False)[0], size)
        l1 = L.upsample(tea.forward(ux1, save=False)[0], size) if mode == 'mix' else None
        kw = dict(loss_fn=loss_fn, conf_thresh=conf_thresh, conf_per_pixel=conf_per_pixel, ramp_val=ramp_val,
                  rampup=rampup, cons_weight=cons_weight)
        x_in = L.paste(ux0, ux1, masks) if mode == 'mix' else ux0 * masks
        lo_mix, sv_mix = stu.forward(x_in)
        lo_mix = leaf(lo_mix)
        if mode == 'mix':
            r = L.mix_mode_loss(L.upsample(lo_mix, size), l0, l1, masks, um0, um1, **kw)
        else:
            r = L.cut_mode_loss(L.upsample(lo_mix, size), l0, masks, um0, **kw)
        total = total + r['unsup_loss']
        closs, rate = r['consistency_loss'], r['conf_rate']
    total.backward()
    if taps_out is not None:       # inputs of every bottleneck + the layer4 output, per student pass (per-block error curves)
        taps_out['sup'] = stu.block_outputs(sv_sup)
        taps_out['sup_logits'] = lo_sup.detach()
        if sv_mix is not None:
            taps_out['mix'] = stu.block_outputs(sv_mix)
            taps_out['mix_logits'] = lo_mix.detach()
    grads = stu.backward(sv_sup, lo_sup.grad)
    if lo_mix is not None and lo_mix.grad is not None:
        stu.backward(sv_mix, lo_mix.grad, grads)
    if grads_out is not None:
        for k in keys:
            grads_out[k] = grads.get(k)
    _apply_updates(S, grads, pi_model)
    return dict(sup_loss=float(sup_loss.detach()),
                consistency_loss=None if closs is None else float(closs.detach()),
                conf_rate=None if rate is None else float(rate))
