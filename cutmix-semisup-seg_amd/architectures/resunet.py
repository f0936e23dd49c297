"""
Mirror of the reference's architectures/resunet.py: U-Net decoder over a torchvision ResNet-50 / ResNet-101 encoder
(BASELINE configs[0]: "resunet on 4 x synthetic 256 x 256 2-class images, supervised-only"), same module tree, attribute
names and state_dict keys (resunet.py:11-116).

  * DecoderBlock (:11-34): nearest x2 upsample, + skip, 3x3 conv (no bias), BatchNorm, ReLU
  * ResUNet.forward (:71-95): taps r2 (after bn1), r4..r32 (layer1..4); line0_conv 2048 -> 1024; decoder3..0; final
    upsample, 3x3 conv, Dropout(0.3), BatchNorm, ReLU, 1x1 classifier
  * quirk kept: `base_model.relu` is an in-place ReLU applied to the very tensor `r2` aliases (:73-74), so the r2 skip
    connection carries relu(bn1(conv1(x))), not the pre-activation value its name suggests
  * BLOCK_SIZE (32, 32), MEAN / STD, pretrained_parameters / new_parameters / freeze_batchnorm (:37-40, 97-108)

Execution: see tv_backbones.py (engine object: batch-statistics BatchNorm on csrc/bn.hip, MFMA convolutions where a layer
fits them, library convolutions otherwise). PARITY UNPINNED (torchvision is absent; checker = oracle/unets.py).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from . import tv_backbones
from .util import freeze_bn_module
from .deeplab3plus import EngineNetMixin


class DecoderBlock(nn.Module):
    def __init__(self, x_chn_in, skip_chn_in, chn_out):
        super(DecoderBlock, self).__init__()
        if x_chn_in != skip_chn_in:
            raise ValueError('x_chn_in != skip_chn_in')
        self.x_chn_in = x_chn_in
        self.skip_chn_in = skip_chn_in
        self.chn_out = chn_out
        self.up = nn.Upsample(scale_factor=2)
        self.conv = nn.Conv2d(x_chn_in, chn_out, 3, padding=1, bias=False)
        self.conv_bn = nn.BatchNorm2d(chn_out)

    def forward(self, x_in, skip_in, eng):
        if x_in.shape[1] != self.x_chn_in:
            raise ValueError('x_in.shape[1]={}, self.x_chn_in={}'.format(x_in.shape[1], self.x_chn_in))
        if skip_in.shape[1] != self.skip_chn_in:
            raise ValueError('skip_in.shape[1]={}, self.skip_chn_in={}'.format(skip_in.shape[1], self.skip_chn_in))
        x = F.interpolate(x_in, scale_factor=2, mode='nearest') + skip_in
        return eng.conv_bn_act(x, self.conv, self.conv_bn, relu=True)


def unet_tail(net, x, eng):
    """final_dec_up -> final_dec_conv -> Dropout(0.3) -> BatchNorm -> ReLU -> final_clf (resunet.py:90-93,
    denseunet.py:127-130)."""
    x = eng.conv2d(F.interpolate(x, scale_factor=2, mode='nearest'), net.final_dec_conv)
    x = F.dropout(x, net.final_dec_drop.p, net.final_dec_drop.training)
    x = eng.bn_act(x, net.final_dec_bn, relu=True)
    clf = net.final_clf
    hip_clf = getattr(eng, 'classifier', None)
    if hip_clf is None:
        raise RuntimeError('engine {} has no classifier kernel for {}'.format(type(eng).__name__, clf))
    return hip_clf(x, clf)                                      # MFMA kernel, fp32 NCHW logits from the epilogue


class ResUNet(EngineNetMixin, nn.Module):
    BLOCK_SIZE = (32, 32)
    MEAN = np.array([0.485, 0.456, 0.406])
    STD = np.array([0.229, 0.224, 0.225])
    upsample_align_corners = True          # (the logits already are at the input resolution)

    def __init__(self, base_model, num_classes, pretrained):
        super(ResUNet, self).__init__()
        self._init_runtime()
        self.base_model = base_model
        self.pretrained = pretrained
        self.line0_conv = nn.Conv2d(2048, 1024, 1)
        self.decoder3 = DecoderBlock(1024, 1024, 512)
        self.decoder2 = DecoderBlock(512, 512, 256)
        self.decoder1 = DecoderBlock(256, 256, 64)
        self.decoder0 = DecoderBlock(64, 64, 64)
        self.final_dec_up = nn.Upsample(scale_factor=2)
        self.final_dec_conv = nn.Conv2d(64, 64, 3, padding=1, bias=False)
        self.final_dec_drop = nn.Dropout(0.3)
        self.final_dec_bn = nn.BatchNorm2d(64)
        self.final_clf = nn.Conv2d(64, num_classes, 1)

    def forward_lowres(self, x):
        """The U-Nets predict at the input resolution: "low-res" logits == logits (fp32)."""
        eng = self._engine(x)
        bm = self.base_model
        x = eng.prepare_input(x)
        r2 = x = eng.conv_bn_act(x, bm.conv1, bm.bn1, relu=True)      # in-place ReLU quirk: r2 IS the activated tensor
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        taps = []
        for layer in (bm.layer1, bm.layer2, bm.layer3, bm.layer4):
            for blk in layer:
                x = blk(x, eng)
            taps.append(x)
        r4, r8, r16, _ = taps
        l0 = self.line0_conv
        x = eng.conv2d(x, l0) + l0.bias.to(x.dtype).view(1, -1, 1, 1)
        x = self.decoder3(x, r16, eng)
        x = self.decoder2(x, r8, eng)
        x = self.decoder1(x, r4, eng)
        x = self.decoder0(x, r2, eng)
        return unet_tail(self, x, eng)

    def forward(self, x):
        return self.forward_lowres(x)

    def pretrained_parameters(self):
        if self.pretrained:
            return list(self.base_model.parameters())
        return []

    def new_parameters(self):
        if self.pretrained:
            pretrained_ids = [id(p) for p in self.base_model.parameters()]
            return [p for p in self.parameters() if id(p) not in pretrained_ids]
        return list(self.parameters())

    def freeze_batchnorm(self):
        self.base_model.apply(freeze_bn_module)


def _no_download(what):
    raise NotImplementedError('pretrained ImageNet weights for {} cannot be downloaded here (no network); build with '
                              'pretrained=False and load a state dict (keys "base_model.*")'.format(what))


def resnet50unet(num_classes, pretrained=True):
    if pretrained:
        _no_download('the torchvision ResNet-50')
    return ResUNet(tv_backbones.resnet50(), num_classes, pretrained=pretrained)


def resnet101unet(num_classes, pretrained=True):
    if pretrained:
        _no_download('the torchvision ResNet-101')
    return ResUNet(tv_backbones.resnet101(), num_classes, pretrained=pretrained)


for _cls in (DecoderBlock, ResUNet):
    _cls.__module__ = 'architectures.resunet'
