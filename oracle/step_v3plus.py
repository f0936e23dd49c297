"""
Oracle (test infrastructure): one CutMix mean-teacher iteration with the DeepLab v3+ network (SURVEY.md 8(a) A4,
BASELINE configs[3]) on PyTorch-CPU fp32 -- train_seg_semisup_mask_mt.py:287-467 in the reference's ORDER, because
this network's head keeps batch-statistics BatchNorm under --freeze_bn (deeplab3plus.py:120-121), so passes cannot be
merged and every forward pass in train mode moves the running statistics of the network it ran on:

    student(x_sup)  -> CE -> backward                         :296-301     (student head stats updated)
    teacher(x0), teacher(x1) under no_grad, TRAIN mode        :354-356     (teacher head stats updated twice, Q4)
    student(x_mix) -> consistency -> backward                 :358-459     (student head stats updated again)
    Adam on every parameter at the full learning rate         :465         (pretrained_parameters() == [], :138-151)
    EMA over every float state tensor incl. running stats     :466-467

Dropout(0.5) of the ASPP projection is random in the reference; the oracle takes it as OFF (tests set p = 0 on the
device side too). PARITY UNPINNED like oracle/deeplab3plus.py (torchvision arithmetic, no reference vectors).
"""
from collections import OrderedDict

import torch

from . import deeplab3plus as v3
from . import losses as L
from .step import _adam_k


class StepStateV3Plus(object):
    def __init__(self, student_state, num_classes, layers=v3.LAYERS, lr=1e-4, teacher_alpha=0.99):
        self.layers = tuple(layers)
        self.num_classes = num_classes
        self.student = OrderedDict((k, v.clone()) for k, v in student_state.items())
        self.teacher = OrderedDict((k, v.clone()) for k, v in student_state.items())
        self.teacher_alpha = teacher_alpha
        self.lr = lr
        self.keys = v3.trainable_keys(num_classes, layers)
        self.m = {k: torch.zeros_like(self.student[k]) for k in self.keys}
        self.v = {k: torch.zeros_like(self.student[k]) for k in self.keys}
        self.steps = {k: 0 for k in self.keys}


def _fwd_train(x, st, layers, commit_to):
    """Train-mode forward with a frozen backbone; the head's new running statistics are written into `commit_to`."""
    ns = {}
    y = v3.forward(x, st, layers, backbone_frozen=True, head_frozen=False, new_stats=ns)
    with torch.no_grad():
        for k, v in ns.items():
            commit_to[k].copy_(v)
    return y


def train_iteration(S, sup_x, sup_y, ux0, ux1, um0, um1, masks, loss_fn='var', conf_thresh=0.97,
                    conf_per_pixel=False, ramp_val=1.0, rampup=-1, cons_weight=1.0):
    leaves = {k: S.student[k].clone().requires_grad_(True) for k in S.keys}
    st = OrderedDict(S.student)
    st.update(leaves)

    sup_loss = L.supervised_ce(_fwd_train(sup_x, st, S.layers, S.student), sup_y[:, 0])
    sup_loss.backward()
    with torch.no_grad():
        l0 = _fwd_train(ux0, S.teacher, S.layers, S.teacher)
        l1 = _fwd_train(ux1, S.teacher, S.layers, S.teacher)
    st = OrderedDict(S.student)          # running statistics moved; leaves stay
    st.update(leaves)
    l_stu = _fwd_train(L.paste(ux0, ux1, masks), st, S.layers, S.student)
    r = L.mix_mode_loss(l_stu, l0, l1, masks, um0, um1, loss_fn=loss_fn, conf_thresh=conf_thresh,
                        conf_per_pixel=conf_per_pixel, ramp_val=ramp_val, rampup=rampup, cons_weight=cons_weight)
    r['unsup_loss'].backward()

    S.last_grads = {k: (None if leaves[k].grad is None else leaves[k].grad.clone()) for k in S.keys}
    with torch.no_grad():
        for k in S.keys:
            g = leaves[k].grad
            if g is None:
                continue
            S.steps[k] = _adam_k(S.student[k], g, S.m[k], S.v[k], S.steps[k], S.lr, 1)
        a = S.teacher_alpha
        for k, t in S.teacher.items():
            if t.dtype == torch.float32:
                t.mul_(a)
                t.add_(S.student[k] * (1.0 - a))
    return dict(sup_loss=float(sup_loss.detach()), consistency_loss=float(r['consistency_loss'].detach()),
                conf_rate=None if r['conf_rate'] is None else float(r['conf_rate']))
