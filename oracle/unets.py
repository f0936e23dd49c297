"""
Oracle (test infrastructure): the reference's U-Nets as *functional* fp32 PyTorch-CPU restatements driven by a plain
state dict with the reference's key names.

  resunet_forward()    -- architectures/resunet.py:71-95 over torchvision's ResNet-50 / 101 (v1.5 bottlenecks: stride on
                          the 3x3; stem max-pool 3x3/2 pad 1 WITHOUT ceil_mode), incl. the in-place-ReLU aliasing of the
                          r2 skip connection (:73-74)
  denseunet_forward()  -- architectures/denseunet.py:105-132 over torchvision's DenseNet-161 (growth 48, blocks
                          6/12/36/24, 96 initial features, bn_size 4; BN-ReLU-conv ordering, 2x2 average-pool transitions)

PARITY UNPINNED: torchvision 0.5.0 (environment.yml:139) is a third-party dependency absent from /root/reference and from
this image, and the reference holds no vectors for these networks. The encoders are restated from torchvision's
published structure; what pins them here is structural: parameter counts equal the published ones (ResNet-50
25 557 032, ResNet-101 44 549 160, DenseNet-161 28 681 000 -- tests/test_host_api.py) and the state_dict key lists follow
torchvision's naming. This file and the product modules (cutmix-semisup-seg_amd/architectures/{tv_backbones,resunet,
denseunet}.py) are two independent statements checked against each other.

BatchNorm: `train=True` = batch statistics (the reference trains these networks without --freeze_bn,
run_isic2017_experiments.sh:14); dropout is left out (p = 0.3 layers are compared in eval mode).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _bn(x, st, pre, train):
    w, b = st[pre + '.weight'], st[pre + '.bias']
    if train:
        return F.batch_norm(x, None, None, w, b, True, 0.0, BN_EPS)
    return F.batch_norm(x, st[pre + '.running_mean'], st[pre + '.running_var'], w, b, False, 0.0, BN_EPS)


def _bottleneck(x, st, pre, stride, has_down, train):
    out = F.relu(_bn(F.conv2d(x, st[pre + '.conv1.weight']), st, pre + '.bn1', train))
    out = F.relu(_bn(F.conv2d(out, st[pre + '.conv2.weight'], stride=stride, padding=1), st, pre + '.bn2', train))
    out = _bn(F.conv2d(out, st[pre + '.conv3.weight']), st, pre + '.bn3', train)
    res = x
    if has_down:
        res = _bn(F.conv2d(x, st[pre + '.downsample.0.weight'], stride=stride), st, pre + '.downsample.1', train)
    return F.relu(out + res)


def _decoder(x, skip, st, pre, train):
    x = F.interpolate(x, scale_factor=2, mode='nearest') + skip
    return F.relu(_bn(F.conv2d(x, st[pre + '.conv.weight'], padding=1), st, pre + '.conv_bn', train))


def _tail(x, st, train):
    x = F.conv2d(F.interpolate(x, scale_factor=2, mode='nearest'), st['final_dec_conv.weight'], padding=1)
    x = F.relu(_bn(x, st, 'final_dec_bn', train))
    return F.conv2d(x, st['final_clf.weight'], st['final_clf.bias'])


def resunet_forward(x, st, layers, train=True):
    p = 'base_model.'
    x = F.relu(_bn(F.conv2d(x, st[p + 'conv1.weight'], stride=2, padding=3), st, p + 'bn1', train))
    r2 = x                                            # (already activated: in-place ReLU on the aliased tensor)
    x = F.max_pool2d(x, 3, 2, 1)
    taps = []
    for li, nblk in enumerate(layers):
        for b in range(nblk):
            x = _bottleneck(x, st, '{}layer{}.{}'.format(p, li + 1, b), 2 if (b == 0 and li > 0) else 1, b == 0, train)
        taps.append(x)
    r4, r8, r16, _ = taps
    x = F.conv2d(x, st['line0_conv.weight'], st['line0_conv.bias'])
    x = _decoder(x, r16, st, 'decoder3', train)
    x = _decoder(x, r8, st, 'decoder2', train)
    x = _decoder(x, r4, st, 'decoder1', train)
    x = _decoder(x, r2, st, 'decoder0', train)
    return _tail(x, st, train)


def denseunet_forward(x, st, block_config=(6, 12, 36, 24), train=True):
    f = 'base_model.features.'
    x = F.relu(_bn(F.conv2d(x, st[f + 'conv0.weight'], stride=2, padding=3), st, f + 'norm0', train))
    enc = [x]                                         # tap 'pool0'
    x = F.max_pool2d(x, 3, 2, 1)
    for bi, nl in enumerate(block_config):
        feats = [x]
        for li in range(nl):
            pre = '{}denseblock{}.denselayer{}'.format(f, bi + 1, li + 1)
            y = torch.cat(feats, 1)
            y = F.conv2d(F.relu(_bn(y, st, pre + '.norm1', train)), st[pre + '.conv1.weight'])
            y = F.conv2d(F.relu(_bn(y, st, pre + '.norm2', train)), st[pre + '.conv2.weight'], padding=1)
            feats.append(y)
        x = torch.cat(feats, 1)
        if bi != len(block_config) - 1:
            enc.append(x)                             # taps 'transition1..3'
            pre = '{}transition{}'.format(f, bi + 1)
            x = F.avg_pool2d(F.conv2d(F.relu(_bn(x, st, pre + '.norm', train)), st[pre + '.conv.weight']), 2, 2)
    x = F.relu(_bn(x, st, f + 'norm5', train))
    enc[-1] = F.conv2d(enc[-1], st['line0_conv.weight'], st['line0_conv.bias'])
    n_dec = len(enc)
    for i, ex in enumerate(enc[::-1]):
        x = _decoder(x, ex, st, 'decoder_blocks.{}'.format(n_dec - 1 - i), train)
    return _tail(x, st, train)
