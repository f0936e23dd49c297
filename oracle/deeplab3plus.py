"""
Oracle (test infrastructure): DeepLab v3+ (ResNet-101, output stride 8) forward as a functional fp32 PyTorch-CPU
restatement driven by a plain state dict.

PARITY UNPINNED. The arithmetic of this model lives in torchvision 0.5.0 (the reference's environment.yml:139),
which is absent from /root/reference and from this image; the reference holds no test vectors for it. What is
restated here is torchvision 0.5.0's published structure, anchored on the reference's call sites:

  backbone      architectures/deeplab3plus.py:81-101   resnet.resnet101(replace_stride_with_dilation=[F, T, T]) wrapped
                                                       in IntermediateLayerGetter{layer4: 'out', layer1: 'low_level'}
                torchvision ResNet v1.5 Bottleneck: 1x1 -> 3x3 (STRIDE AND DILATION HERE) -> 1x1 (x4), shortcut
                1x1(stride)+BN on the first block of a layer; `_make_layer(dilate=True)` turns the layer's stride into
                dilation: the first block keeps the previous dilation, the others use the new one
                (layer3: 1,2,2,...; layer4: 2,4,4); stem 7x7/2 + BN + ReLU + max-pool 3x3/2 pad 1 (no ceil_mode)
  ASPP          torchvision.models.segmentation.deeplabv3.ASPP(2048, [12, 24, 36]): 1x1 | 3 x (3x3 dilated) | global
                average pool -> 1x1 -> bilinear(align_corners=False) back; each conv(bias=False)+BN+ReLU; concat 1280
                -> 1x1 -> BN -> ReLU -> Dropout(0.5)
  head          DeepLabHeadV3Plus.forward, deeplab3plus.py:50-56: low-level 1x1 256->48 +BN+ReLU; ASPP output
                upsampled (align_corners=False) to the low-level size; concat (48 + 256); 3x3+BN+ReLU, 3x3+BN+ReLU,
                1x1 (+bias) -> classes
  forward       DeepLabV3Plus.forward, deeplab3plus.py:73-78: bilinear to the input size, align_corners=False
  BN modes      DeepLabv3Wrapper.freeze_batchnorm, deeplab3plus.py:120-121: BACKBONE only; all head BatchNorms keep
                batch statistics in train mode, Dropout stays active (also in the teacher)

State-dict keys are those of the reference's wrapper: 'deeplab.backbone.*' / 'deeplab.classifier.*'.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .deeplab2 import BN_EPS, BN_MOMENTUM, _key_seed

LAYERS = (3, 4, 23, 3)
PLANES = (64, 128, 256, 512)
ASPP_RATES = (12, 24, 36)
ASPP_OUT = 256
LOW_LEVEL_OUT = 48
B = 'deeplab.backbone.'
H = 'deeplab.classifier.'


def layer_plan(layers=LAYERS):
    """[(prefix, inplanes, planes, stride, dilation, has_downsample)] for every bottleneck, torchvision order."""
    plan = []
    inplanes, dilation = 64, 1
    for li, (nblk, planes) in enumerate(zip(layers, PLANES)):
        stride = 1 if li == 0 else 2
        prev_dilation = dilation
        if li >= 2:                       # replace_stride_with_dilation = [False, True, True]
            dilation *= stride
            stride = 1
        for b in range(nblk):
            pre = '{}layer{}.{}'.format(B, li + 1, b)
            if b == 0:
                plan.append((pre, inplanes, planes, stride, prev_dilation, stride != 1 or inplanes != planes * 4))
                inplanes = planes * 4
            else:
                plan.append((pre, inplanes, planes, 1, dilation, False))
    return plan


def state_spec(num_classes, layers=LAYERS):
    spec = OrderedDict()

    def bn(prefix, c):
        spec[prefix + '.weight'] = ((c,), torch.float32)
        spec[prefix + '.bias'] = ((c,), torch.float32)
        spec[prefix + '.running_mean'] = ((c,), torch.float32)
        spec[prefix + '.running_var'] = ((c,), torch.float32)
        spec[prefix + '.num_batches_tracked'] = ((), torch.int64)

    spec[B + 'conv1.weight'] = ((64, 3, 7, 7), torch.float32)
    bn(B + 'bn1', 64)
    for pre, inplanes, planes, stride, dil, down in layer_plan(layers):
        spec[pre + '.conv1.weight'] = ((planes, inplanes, 1, 1), torch.float32)
        bn(pre + '.bn1', planes)
        spec[pre + '.conv2.weight'] = ((planes, planes, 3, 3), torch.float32)
        bn(pre + '.bn2', planes)
        spec[pre + '.conv3.weight'] = ((planes * 4, planes, 1, 1), torch.float32)
        bn(pre + '.bn3', planes * 4)
        if down:
            spec[pre + '.downsample.0.weight'] = ((planes * 4, inplanes, 1, 1), torch.float32)
            bn(pre + '.downsample.1', planes * 4)
    spec[H + 'project.0.weight'] = ((LOW_LEVEL_OUT, 256, 1, 1), torch.float32)
    bn(H + 'project.1', LOW_LEVEL_OUT)
    spec[H + 'aspp.convs.0.0.weight'] = ((ASPP_OUT, 2048, 1, 1), torch.float32)
    bn(H + 'aspp.convs.0.1', ASPP_OUT)
    for i in range(3):
        spec[H + 'aspp.convs.{}.0.weight'.format(i + 1)] = ((ASPP_OUT, 2048, 3, 3), torch.float32)
        bn(H + 'aspp.convs.{}.1'.format(i + 1), ASPP_OUT)
    spec[H + 'aspp.convs.4.1.weight'] = ((ASPP_OUT, 2048, 1, 1), torch.float32)     # index 0 is the pooling module
    bn(H + 'aspp.convs.4.2', ASPP_OUT)
    spec[H + 'aspp.project.0.weight'] = ((ASPP_OUT, 5 * ASPP_OUT, 1, 1), torch.float32)
    bn(H + 'aspp.project.1', ASPP_OUT)
    spec[H + 'classifier.0.weight'] = ((256, LOW_LEVEL_OUT + ASPP_OUT, 3, 3), torch.float32)
    bn(H + 'classifier.1', 256)
    spec[H + 'classifier.3.weight'] = ((256, 256, 3, 3), torch.float32)
    bn(H + 'classifier.4', 256)
    spec[H + 'classifier.6.weight'] = ((num_classes, 256, 1, 1), torch.float32)
    spec[H + 'classifier.6.bias'] = ((num_classes,), torch.float32)
    return spec


def closed_form_state(num_classes, layers=LAYERS, dtype=torch.float32):
    """Deterministic O(1)-activation weights, same recipe as oracle.deeplab2.closed_form_state."""
    out = OrderedDict()
    for key, (shape, dt) in state_spec(num_classes, layers).items():
        if dt == torch.int64:
            out[key] = torch.zeros(shape, dtype=torch.int64)
            continue
        n = int(np.prod(shape)) if len(shape) else 1
        s = np.sin((_key_seed(key) % 10007) * 0.001 + 0.7548776662 * np.arange(n, dtype=np.float64))
        if key.endswith('running_var'):
            v = 1.0 + 0.2 * s
        elif key.endswith('running_mean'):
            v = 0.05 * s
        elif len(shape) == 1 and key.endswith('weight'):
            v = 1.0 + 0.2 * s                           # BatchNorm gamma
        elif key.endswith('bias'):
            v = 0.1 * s
        else:
            v = s * np.sqrt(3.0 / int(np.prod(shape[1:])))
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


def backbone(x, st, layers=LAYERS, frozen=True, new_stats=None):
    """-> (low_level = layer1 output, out = layer4 output)."""
    x = F.conv2d(x, st[B + 'conv1.weight'], stride=2, padding=3)
    x = F.relu(_bn(x, st, B + 'bn1', frozen, new_stats))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    low = None
    n1 = layers[0]
    for i, (pre, inplanes, planes, stride, dil, down) in enumerate(layer_plan(layers)):
        out = F.relu(_bn(F.conv2d(x, st[pre + '.conv1.weight']), st, pre + '.bn1', frozen, new_stats))
        out = F.conv2d(out, st[pre + '.conv2.weight'], stride=stride, padding=dil, dilation=dil)
        out = F.relu(_bn(out, st, pre + '.bn2', frozen, new_stats))
        out = _bn(F.conv2d(out, st[pre + '.conv3.weight']), st, pre + '.bn3', frozen, new_stats)
        res = x
        if down:
            res = _bn(F.conv2d(x, st[pre + '.downsample.0.weight'], stride=stride), st, pre + '.downsample.1',
                      frozen, new_stats)
        x = F.relu(out + res)
        if i == n1 - 1:
            low = x
    return low, x


def aspp(x, st, head_frozen, new_stats=None, drop_mask=None):
    """drop_mask: None = no dropout (eval); else a {0,1} tensor of the ASPP output's shape, applied with 1/(1-p), p=0.5."""
    pre = H + 'aspp.'
    br = [F.relu(_bn(F.conv2d(x, st[pre + 'convs.0.0.weight']), st, pre + 'convs.0.1', head_frozen, new_stats))]
    for i, r in enumerate(ASPP_RATES):
        y = F.conv2d(x, st[pre + 'convs.{}.0.weight'.format(i + 1)], padding=r, dilation=r)
        br.append(F.relu(_bn(y, st, pre + 'convs.{}.1'.format(i + 1), head_frozen, new_stats)))
    g = x.mean(dim=(2, 3), keepdim=True)
    g = F.relu(_bn(F.conv2d(g, st[pre + 'convs.4.1.weight']), st, pre + 'convs.4.2', head_frozen, new_stats))
    br.append(F.interpolate(g, size=x.shape[2:4], mode='bilinear', align_corners=False))
    y = F.conv2d(torch.cat(br, dim=1), st[pre + 'project.0.weight'])
    y = F.relu(_bn(y, st, pre + 'project.1', head_frozen, new_stats))
    if drop_mask is not None:
        y = y * drop_mask * 2.0
    return y


def head(low, out, st, head_frozen, new_stats=None, drop_mask=None):
    lo = F.relu(_bn(F.conv2d(low, st[H + 'project.0.weight']), st, H + 'project.1', head_frozen, new_stats))
    a = aspp(out, st, head_frozen, new_stats, drop_mask)
    a = F.interpolate(a, size=lo.shape[2:4], mode='bilinear', align_corners=False)
    y = torch.cat([lo, a], dim=1)
    y = F.relu(_bn(F.conv2d(y, st[H + 'classifier.0.weight'], padding=1), st, H + 'classifier.1', head_frozen, new_stats))
    y = F.relu(_bn(F.conv2d(y, st[H + 'classifier.3.weight'], padding=1), st, H + 'classifier.4', head_frozen, new_stats))
    return F.conv2d(y, st[H + 'classifier.6.weight'], st[H + 'classifier.6.bias'])


def forward_lowres(x, st, layers=LAYERS, backbone_frozen=True, head_frozen=True, new_stats=None, drop_mask=None):
    low, out = backbone(x, st, layers, backbone_frozen, new_stats)
    return head(low, out, st, head_frozen, new_stats, drop_mask)


def forward(x, st, layers=LAYERS, backbone_frozen=True, head_frozen=True, new_stats=None, drop_mask=None):
    lo = forward_lowres(x, st, layers, backbone_frozen, head_frozen, new_stats, drop_mask)
    return F.interpolate(lo, size=x.shape[2:4], mode='bilinear', align_corners=False)


def trainable_keys(num_classes, layers=LAYERS):
    """torchvision leaves every parameter trainable (BN affine included) -- unlike DeepLab v2."""
    return [k for k, (shape, dt) in state_spec(num_classes, layers).items()
            if dt == torch.float32 and not k.endswith('running_mean') and not k.endswith('running_var')]
