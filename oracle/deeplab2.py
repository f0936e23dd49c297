"""
Oracle (test infrastructure): DeepLab v2 (ResNet-101) forward as a *functional* fp32 PyTorch-CPU restatement,
driven by a plain state dict with the reference's key names.

  forward_lowres()   -- architectures/deeplab2.py:183-193  conv1/bn1/relu/maxpool(ceil) -> layer1..4 -> layer5
  bottleneck()       -- architectures/deeplab2.py:89-109   (stride on the first 1x1, :70; dilated 3x3, :76-77)
  aspp_head()        -- architectures/deeplab2.py:124-128  early `return` inside the loop => only dilations
                        6 and 12 contribute
  forward()          -- + bilinear upsample align_corners=True, architectures/deeplab2.py:204
  state_spec()       -- names/shapes of the 632 state tensors (deeplab2.py:131-181)
  param_multiplicity() -- how many times each trainable tensor is yielded by pretrained_parameters()
                        (deeplab2.py:208-230) and new_parameters() (:232-242): SURVEY.md 8(a) A3b
  closed_form_state()  -- deterministic weights as a function of (key, flat index); lets golden outputs be
                        committed without shipping a weight file

BatchNorm: `frozen=True` uses running statistics (what `--freeze_bn` + `freeze_batchnorm()` gives,
architectures/util.py:2-10); `frozen=False` uses batch statistics and returns updated running stats like
nn.BatchNorm2d in train mode (momentum 0.1, unbiased running variance).

Pinned by tests/golden/deeplab2_*.npz (outputs of the reference nn.Module on the same closed-form weights).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = (3, 4, 23, 3)
PLANES = (64, 128, 256, 512)
STRIDES = (1, 2, 1, 1)
DILATIONS = (1, 1, 2, 4)
ASPP_DILATIONS = (6, 12, 18, 24)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def state_spec(num_classes, layers=LAYERS):
    """OrderedDict name -> (shape, dtype) in nn.Module.state_dict() order."""
    spec = OrderedDict()

    def bn(prefix, c):
        spec[prefix + '.weight'] = ((c,), torch.float32)
        spec[prefix + '.bias'] = ((c,), torch.float32)
        spec[prefix + '.running_mean'] = ((c,), torch.float32)
        spec[prefix + '.running_var'] = ((c,), torch.float32)
        spec[prefix + '.num_batches_tracked'] = ((), torch.int64)

    spec['conv1.weight'] = ((64, 3, 7, 7), torch.float32)
    bn('bn1', 64)
    inplanes = 64
    for li, (nblk, planes) in enumerate(zip(layers, PLANES)):
        for b in range(nblk):
            pre = 'layer{}.{}'.format(li + 1, b)
            spec[pre + '.conv1.weight'] = ((planes, inplanes, 1, 1), torch.float32)
            bn(pre + '.bn1', planes)
            spec[pre + '.conv2.weight'] = ((planes, planes, 3, 3), torch.float32)
            bn(pre + '.bn2', planes)
            spec[pre + '.conv3.weight'] = ((planes * 4, planes, 1, 1), torch.float32)
            bn(pre + '.bn3', planes * 4)
            if b == 0:
                # deeplab2.py:163-168: every first block has a downsample branch (channel change, stride
                # or dilation 2/4)
                spec[pre + '.downsample.0.weight'] = ((planes * 4, inplanes, 1, 1), torch.float32)
                bn(pre + '.downsample.1', planes * 4)
                inplanes = planes * 4
    for i in range(4):
        spec['layer5.conv2d_list.{}.weight'.format(i)] = ((num_classes, 2048, 3, 3), torch.float32)
        spec['layer5.conv2d_list.{}.bias'.format(i)] = ((num_classes,), torch.float32)
    return spec


def _key_seed(key):
    h = 2166136261
    for ch in key.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def closed_form_state(num_classes, layers=LAYERS, dtype=torch.float32):
    """
    Deterministic, non-degenerate weights: conv ~ amplitude chosen to keep activations O(1) through 100 layers,
    BN gamma in [0.8, 1.2], beta in [-0.1, 0.1], running_mean in [-0.05, 0.05], running_var in [0.8, 1.2].
    value(key, i) = f(sin(phase(key) + 0.7548776662 * i)).
    """
    out = OrderedDict()
    for key, (shape, dt) in state_spec(num_classes, layers).items():
        if dt == torch.int64:
            out[key] = torch.zeros(shape, dtype=torch.int64)
            continue
        n = int(np.prod(shape)) if len(shape) else 1
        idx = np.arange(n, dtype=np.float64)
        phase = (_key_seed(key) % 10007) * 0.001
        s = np.sin(phase + 0.7548776662 * idx)
        if key.endswith('running_var'):
            v = 1.0 + 0.2 * s
        elif key.endswith('running_mean'):
            v = 0.05 * s
        elif '.bn' in key or key.startswith('bn1') or 'downsample.1' in key:
            v = (1.0 + 0.2 * s) if key.endswith('weight') else 0.1 * s
        elif key.endswith('bias'):
            v = 0.1 * s
        else:
            fan_in = int(np.prod(shape[1:]))
            v = s * math.sqrt(3.0 / fan_in)
        out[key] = torch.tensor(v.reshape(shape), dtype=dtype)
    return out


def _bn(x, st, prefix, frozen, new_stats):
    w, b = st[prefix + '.weight'], st[prefix + '.bias']
    rm, rv = st[prefix + '.running_mean'], st[prefix + '.running_var']
    if frozen:
        return F.batch_norm(x, rm, rv, w, b, False, 0.0, BN_EPS)
    rm2, rv2 = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm2, rv2, w, b, True, BN_MOMENTUM, BN_EPS)
    if new_stats is not None:
        new_stats[prefix + '.running_mean'] = rm2
        new_stats[prefix + '.running_var'] = rv2
    return y


def bottleneck(x, st, pre, stride, dilation, has_down, frozen, new_stats):
    out = F.conv2d(x, st[pre + '.conv1.weight'], stride=stride)
    out = F.relu(_bn(out, st, pre + '.bn1', frozen, new_stats))
    out = F.conv2d(out, st[pre + '.conv2.weight'], padding=dilation, dilation=dilation)
    out = F.relu(_bn(out, st, pre + '.bn2', frozen, new_stats))
    out = F.conv2d(out, st[pre + '.conv3.weight'])
    out = _bn(out, st, pre + '.bn3', frozen, new_stats)
    if has_down:
        res = F.conv2d(x, st[pre + '.downsample.0.weight'], stride=stride)
        res = _bn(res, st, pre + '.downsample.1', frozen, new_stats)
    else:
        res = x
    return F.relu(out + res)


def aspp_head(x, st):
    out = None
    for i, d in enumerate(ASPP_DILATIONS[:2]):          # only the first two branches are ever summed
        y = F.conv2d(x, st['layer5.conv2d_list.{}.weight'.format(i)], st['layer5.conv2d_list.{}.bias'.format(i)],
                     padding=d, dilation=d)
        out = y if out is None else out + y
    return out


def backbone(x, st, layers=LAYERS, frozen=True, new_stats=None, taps=None):
    x = F.conv2d(x, st['conv1.weight'], stride=2, padding=3)
    x = F.relu(_bn(x, st, 'bn1', frozen, new_stats))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1, ceil_mode=True)
    if taps is not None:
        taps['stem'] = x
    for li, nblk in enumerate(layers):
        for b in range(nblk):
            pre = 'layer{}.{}'.format(li + 1, b)
            x = bottleneck(x, st, pre, STRIDES[li] if b == 0 else 1, DILATIONS[li], b == 0, frozen, new_stats)
        if taps is not None:
            taps['layer{}'.format(li + 1)] = x
    return x


def forward_lowres(x, st, layers=LAYERS, frozen=True, new_stats=None, taps=None):
    return aspp_head(backbone(x, st, layers, frozen, new_stats, taps), st)


def forward(x, st, layers=LAYERS, frozen=True, new_stats=None, taps=None):
    lo = forward_lowres(x, st, layers, frozen, new_stats, taps)
    return F.interpolate(lo, size=x.shape[2:4], mode='bilinear', align_corners=True)


def trainable_keys(num_classes, layers=LAYERS):
    """Conv weights + ASPP weights/biases; every BN affine parameter is requires_grad=False (deeplab2.py:72-84)."""
    keys = []
    for k, (shape, dt) in state_spec(num_classes, layers).items():
        if dt != torch.float32:
            continue
        if k.startswith('layer5.') or (k.endswith('.weight') and len(shape) == 4):
            keys.append(k)
    return keys


def param_multiplicity(num_classes, layers=LAYERS):
    """
    -> (group0: OrderedDict key -> times yielded by pretrained_parameters(), group1: same for new_parameters()).

    pretrained_parameters() walks `modules()` of [conv1, bn1, layer1..4] and, for each module visited, yields all
    of that module's (recursive) trainable parameters. A conv weight `layerK.B.convJ.weight` is therefore yielded
    once for each ancestor-or-self inside the walked subtree: layerK (Sequential), layerK.B (Bottleneck), the conv
    itself = 3; `layerK.0.downsample.0.weight` additionally via the `downsample` Sequential = 4; `conv1.weight`
    only via itself = 1.
    """
    g0 = OrderedDict()
    for k in trainable_keys(num_classes, layers):
        if k.startswith('layer5.'):
            continue
        if k == 'conv1.weight':
            g0[k] = 1
        elif '.downsample.' in k:
            g0[k] = 4
        else:
            g0[k] = 3
    g1 = OrderedDict((k, 1) for k in trainable_keys(num_classes, layers) if k.startswith('layer5.'))
    return g0, g1


def pretrained_param_order(num_classes, layers=LAYERS):
    """The exact sequence (with repeats) in which pretrained_parameters() yields keys; optimizer entry order."""
    seq = ['conv1.weight']
    inpl_has_down = True
    for li, nblk in enumerate(layers):
        L = 'layer{}'.format(li + 1)

        def block_keys(b):
            ks = ['{}.{}.conv1.weight'.format(L, b), '{}.{}.conv2.weight'.format(L, b),
                  '{}.{}.conv3.weight'.format(L, b)]
            if b == 0:
                ks.append('{}.0.downsample.0.weight'.format(L))
            return ks

        # module = the Sequential itself: all parameters in registration order
        for b in range(nblk):
            seq += block_keys(b)
        # then each descendant module in modules() (pre-order): block, its convs, bns (no trainable params),
        # relu, downsample Sequential, downsample conv, downsample bn
        for b in range(nblk):
            seq += block_keys(b)                                   # the Bottleneck
            seq += ['{}.{}.conv1.weight'.format(L, b)]             # conv1 (bn1 yields nothing)
            seq += ['{}.{}.conv2.weight'.format(L, b)]
            seq += ['{}.{}.conv3.weight'.format(L, b)]
            if b == 0:
                seq += ['{}.0.downsample.0.weight'.format(L)]      # the downsample Sequential
                seq += ['{}.0.downsample.0.weight'.format(L)]      # its conv
    return seq
