"""
Oracle (test infrastructure): DeepLab v3+ as a chain of UNITS -- raw convolution, BatchNorm (+ residual) (+ ReLU) -- each with
its forward AND backward written out and an optional model of bf16 STORAGE: what the bf16 (timed) configuration of the
device's layer engine is held to, unit by unit and teacher-forced (tests/test_gpu_deeplab3plus.py).

Why units. Whole-network comparisons of two bf16 pipelines decorrelate (one flipped bf16 tie perturbs every
later tensor by as much as the storage noise itself), so the tight statement about the timed configuration is per unit, from
the device's OWN stored inputs. The device's layer engine (architectures/deeplab3plus.py: HipConvEngine, the engine of the
v3+ head and -- on batch statistics -- of its backbone) stores, in bf16:

    u   = R(conv(x, R(W)))                                            raw convolution output          deeplab3plus.py:40-56
    y   = R(relu(u * scale + shift (+ res)))                          BatchNorm (+ residual) (+ ReLU)
          batch statistics (the head, always: deeplab3plus.py:120-121): mean / biased variance of u over (N, H, W) per sample
          group, scale = gamma * rstd, shift = beta - mean * scale;   frozen: scale / shift from the running statistics
    dy' = dy * [y > 0]                                                 backward of the ReLU (mask from the stored y)
    du  = R(gamma * rstd * (dy' - mean(dy') - xhat * mean(dy' * xhat)))      (frozen: R(dy' * scale));   dres = R(dy')
    dx  = R(conv^T(du, R(W)))          dW = wgrad(x, du)  (fp32)       dgamma = sum(dy' * xhat), dbeta = sum(dy')

with R = round-to-nearest-even to bf16 (`storage='bf16'`) or the identity (`storage='fp32'`). Everything between two roundings
is fp32 / fp64 PyTorch-CPU (ATen convolutions: the ops the reference calls).

Ties. With storage off, `forward_lowres` below -- the v3+ graph assembled from these units -- equals oracle/deeplab3plus.py
(<= 4e-6), and the unit backward functions equal ATen's autograd of the same unit (tests/test_oracle_chain.py). The graph
itself (torchvision 0.5's ResNet v1.5 + ASPP + the reference's head) is UNPINNED like oracle/deeplab3plus.py: torchvision is
absent from the reference tree and from this image; the rounding points restate this repository's storage format.
"""
import torch
import torch.nn.functional as F
from torch.nn.grad import conv2d_input, conv2d_weight

from . import deeplab3plus as o3
from .deeplab2 import BN_EPS, BN_MOMENTUM


def rb(t, storage):
    return t.bfloat16().float() if storage == 'bf16' else t


# ------------------------------------------------------------------------------------------------------------- units
def conv_unit(x, w, stride=1, padding=0, dilation=1, storage='bf16', bias=None):
    """u = R(conv(R(x), R(W)) (+ bias)): the raw convolution output as stored."""
    u = F.conv2d(rb(x, storage), rb(w, storage), None, stride, padding, dilation)
    if bias is not None:
        u = u + bias.view(1, -1, 1, 1)
    return rb(u, storage)


def conv_unit_backward(x, w, du, stride=1, padding=0, dilation=1, storage='bf16'):
    """(dx, dW): dx = R(conv^T(du, R(W))), dW = wgrad(R(x), du) in fp32."""
    xs, ws = rb(x, storage), rb(w, storage)
    dx = conv2d_input(xs.shape, ws, du, stride, padding, dilation)
    dw = conv2d_weight(xs, ws.shape, du, stride, padding, dilation)
    return rb(dx, storage), dw


def bn_unit(u, gamma, beta, running_mean, running_var, relu, res=None, frozen=False, storage='bf16', groups=1, eps=BN_EPS,
            momentum=BN_MOMENTUM):
    """-> (y, ctx). y = R(relu(u * scale + shift (+ res))); ctx = dict(mean, rstd [per group], new running statistics)."""
    n, c = u.shape[0], u.shape[1]
    if frozen:
        rstd = torch.rsqrt(running_var + eps)
        scale = gamma * rstd
        shift = beta - running_mean * scale
        y = u * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        ctx = dict(mean=running_mean.view(1, c), rstd=rstd.view(1, c), running_mean=running_mean, running_var=running_var, groups=1)
    else:
        ug = u.double().reshape(groups, n // groups, c, -1)
        cnt = ug.shape[1] * ug.shape[3]
        mean = ug.mean(dim=(1, 3))                                           # (G, C)
        var = (ug * ug).mean(dim=(1, 3)) - mean * mean                       # biased: E[x^2] - E[x]^2 like the kernels' sums
        var = var.clamp_min(0.0)
        rstd = (1.0 / torch.sqrt(var + eps)).float()
        mean32 = mean.float()
        scale = gamma.view(1, c) * rstd
        shift = beta.view(1, c) - mean32 * scale
        y = (u.reshape(groups, n // groups, c, -1) * scale.view(groups, 1, c, 1) + shift.view(groups, 1, c, 1)).reshape(u.shape)
        rm, rv = running_mean.clone(), running_var.clone()
        for g in range(groups):                                              # the running statistics move once per group, in order
            unb = var[g] * (cnt / max(cnt - 1.0, 1.0))
            rm = (1.0 - momentum) * rm + momentum * mean32[g]
            rv = (1.0 - momentum) * rv + momentum * unb.float()
        ctx = dict(mean=mean32, rstd=rstd, running_mean=rm, running_var=rv, groups=groups)
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    return rb(y, storage), ctx


def bn_unit_backward(u, y, dy, gamma, ctx, relu, has_res, frozen=False, storage='bf16'):
    """-> (du, dres or None, dgamma, dbeta). `y`: the stored output (its sign is the ReLU mask)."""
    n, c = u.shape[0], u.shape[1]
    g = ctx['groups']
    d = dy * (y > 0).to(dy.dtype) if relu else dy
    mean, rstd = ctx['mean'], ctx['rstd']                                    # (G, C) (frozen: (1, C))
    ug = u.reshape(g, n // g, c, -1)
    dg = d.reshape(g, n // g, c, -1)
    xhat = (ug - mean.view(g, 1, c, 1)) * rstd.view(g, 1, c, 1)
    s0 = dg.double().sum(dim=(1, 3))                                         # (G, C): sum dy'
    s1 = (dg.double() * xhat.double()).sum(dim=(1, 3))                       # sum dy' xhat
    if frozen:
        du = dg * (gamma.view(1, c) * rstd).view(g, 1, c, 1)
    else:
        cnt = float(ug.shape[1] * ug.shape[3])
        du = (gamma.view(1, c) * rstd).view(g, 1, c, 1) * (dg - (s0 / cnt).float().view(g, 1, c, 1)
                                                           - xhat * (s1 / cnt).float().view(g, 1, c, 1))
    du = rb(du.reshape(u.shape), storage)
    dres = rb(d, storage) if has_res else None
    return du, dres, s1.sum(0).float(), s0.sum(0).float()


def fused_frozen_unit(x, w, gamma, beta, running_mean, running_var, stride=1, padding=0, dilation=1, relu=True, res=None,
                      storage='bf16', eps=BN_EPS):
    """The EXECUTOR's unit (backbone on frozen statistics, backbone_hip.py): convolution, folded BatchNorm affine, residual and
    ReLU in ONE epilogue, ONE rounding:  y = R(relu(conv(R(x), R(W)) * scale + shift (+ res)))."""
    scale = gamma * torch.rsqrt(running_var + eps)
    shift = beta - running_mean * scale
    y = F.conv2d(rb(x, storage), rb(w, storage), None, stride, padding, dilation)
    y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    return rb(y, storage)


# ------------------------------------------------------------------------------------------------------------- the graph
def _cba(x, st, conv_key, bn_prefix, stride, padding, dilation, relu, res, frozen, storage, taps, groups=1):
    u = conv_unit(x, st[conv_key], stride, padding, dilation, storage)
    y, ctx = bn_unit(u, st[bn_prefix + '.weight'], st[bn_prefix + '.bias'], st[bn_prefix + '.running_mean'],
                     st[bn_prefix + '.running_var'], relu, res, frozen, storage, groups)
    if taps is not None:
        taps.append((conv_key, x, u, y))
    return y


def forward_lowres(x, st, layers=o3.LAYERS, backbone_frozen=True, head_frozen=True, storage='bf16', taps=None, groups=1):
    """oracle/deeplab3plus.py's forward assembled from the units above (no dropout). `taps`: list that receives
    (convolution key, unit input, raw convolution output, unit output) of every unit."""
    B, H = o3.B, o3.H
    x = rb(x, storage)
    x = _cba(x, st, B + 'conv1.weight', B + 'bn1', 2, 3, 1, True, None, backbone_frozen, storage, taps, groups)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    low = None
    for i, (pre, inplanes, planes, stride, dil, down) in enumerate(o3.layer_plan(layers)):
        a1 = _cba(x, st, pre + '.conv1.weight', pre + '.bn1', 1, 0, 1, True, None, backbone_frozen, storage, taps, groups)
        a2 = _cba(a1, st, pre + '.conv2.weight', pre + '.bn2', stride, dil, dil, True, None, backbone_frozen, storage, taps, groups)
        res = x
        if down:
            res = _cba(x, st, pre + '.downsample.0.weight', pre + '.downsample.1', stride, 0, 1, False, None, backbone_frozen,
                       storage, taps, groups)
        x = _cba(a2, st, pre + '.conv3.weight', pre + '.bn3', 1, 0, 1, True, res, backbone_frozen, storage, taps, groups)
        if i == layers[0] - 1:
            low = x
    out = x
    lo = _cba(low, st, H + 'project.0.weight', H + 'project.1', 1, 0, 1, True, None, head_frozen, storage, taps, groups)
    pre = H + 'aspp.'
    br = [_cba(out, st, pre + 'convs.0.0.weight', pre + 'convs.0.1', 1, 0, 1, True, None, head_frozen, storage, taps, groups)]
    for i, r in enumerate(o3.ASPP_RATES):
        br.append(_cba(out, st, pre + 'convs.{}.0.weight'.format(i + 1), pre + 'convs.{}.1'.format(i + 1), 1, r, r, True, None,
                       head_frozen, storage, taps, groups))
    g = rb(out.mean(dim=(2, 3), keepdim=True), storage)
    g = _cba(g, st, pre + 'convs.4.1.weight', pre + 'convs.4.2', 1, 0, 1, True, None, head_frozen, storage, taps, groups)
    br.append(rb(F.interpolate(g, size=out.shape[2:4], mode='bilinear', align_corners=False), storage))
    y = _cba(torch.cat(br, dim=1), st, pre + 'project.0.weight', pre + 'project.1', 1, 0, 1, True, None, head_frozen, storage,
             taps, groups)
    a = rb(F.interpolate(y, size=lo.shape[2:4], mode='bilinear', align_corners=False), storage)
    y = torch.cat([lo, a], dim=1)
    y = _cba(y, st, H + 'classifier.0.weight', H + 'classifier.1', 1, 1, 1, True, None, head_frozen, storage, taps, groups)
    y = _cba(y, st, H + 'classifier.3.weight', H + 'classifier.4', 1, 1, 1, True, None, head_frozen, storage, taps, groups)
    # the classifier's logits leave the epilogue in fp32 (not rounded)
    return F.conv2d(rb(y, storage), rb(st[H + 'classifier.6.weight'], storage), st[H + 'classifier.6.bias'])
