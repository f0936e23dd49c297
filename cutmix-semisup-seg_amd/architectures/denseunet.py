"""
Mirror of the reference's architectures/denseunet.py: U-Net decoder over a torchvision DenseNet-161 encoder -- the network
of BASELINE configs[4] (ISIC 2017, VAT trainer), same module tree, attribute names and state_dict keys
(denseunet.py:11-153).

  * encoder taps (:56-76): the inputs of pool0 (96 ch, 1/2), transition1 (384, 1/4), transition2 (768, 1/8),
    transition3 (2112, 1/16); line0_conv 2112 -> 2208 on the last tap; ReLU after norm5 (:113)
  * decoder (:78-93): DecoderBlock(2208, 2208, 768), (768, 768, 384), (384, 384, 96), (96, 96, 96), kept in the
    reference's reversed ModuleList order (state_dict keys decoder_blocks.0 = the LAST block applied)
  * tail (:96-103, 127-130): upsample, 3x3 conv 96 -> 64, Dropout(0.3), BatchNorm, ReLU, 1x1 classifier
  * BLOCK_SIZE (32, 32); pretrained_parameters / new_parameters / freeze_batchnorm (:134-143)

Execution: see tv_backbones.py. PARITY UNPINNED (torchvision is absent; checker = oracle/unets.py).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import tv_backbones
from .util import freeze_bn_module
from .deeplab3plus import EngineNetMixin
from .resunet import DecoderBlock as _ResDecoderBlock, unet_tail


class DecoderBlock(_ResDecoderBlock):
    pass


class DenseUNet(EngineNetMixin, nn.Module):
    BLOCK_SIZE = (32, 32)
    MEAN = np.array([0.485, 0.456, 0.406])
    STD = np.array([0.229, 0.224, 0.225])
    upsample_align_corners = True

    def __init__(self, base_model, num_classes, mean, std, pretrained):
        super(DenseUNet, self).__init__()
        self._init_runtime()
        self.MEAN = mean
        self.STD = std
        self.pretrained = pretrained
        self.tap_names = ['pool0', 'transition1', 'transition2', 'transition3']
        self.base_model = base_model
        f = base_model.features
        enc_chn = [f.norm0.num_features, f.transition1.norm.num_features, f.transition2.norm.num_features,
                   f.transition3.norm.num_features]
        n_chn = f.norm5.num_features
        self.line0_conv = nn.Conv2d(enc_chn[-1], n_chn, 1)
        enc_chn[-1] = n_chn
        blocks = []
        enc_chn = enc_chn[::-1]
        for e_chn_a, e_chn_b in zip(enc_chn, enc_chn[1:] + enc_chn[-1:]):
            blocks.append(DecoderBlock(n_chn, e_chn_a, e_chn_b))
            n_chn = e_chn_b
        self.decoder_blocks = nn.ModuleList(blocks[::-1])
        self.final_dec_up = nn.Upsample(scale_factor=2)
        self.final_dec_conv = nn.Conv2d(n_chn, 64, 3, padding=1, bias=False)
        self.final_dec_drop = nn.Dropout(0.3)
        self.final_dec_bn = nn.BatchNorm2d(64)
        self.final_clf = nn.Conv2d(64, num_classes, 1)

    def forward_lowres(self, x):
        eng = self._engine(x)
        f = self.base_model.features
        x = eng.prepare_input(x)
        enc_x = []
        x = eng.conv_bn_act(x, f.conv0, f.norm0, relu=True)
        for name, mod in f.named_children():
            if name in ('conv0', 'norm0', 'relu0'):
                continue
            if name in self.tap_names:
                enc_x.append(x)
            if name == 'pool0':
                x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
            elif name == 'norm5':
                x = eng.bn_act(x, mod, relu=True)                 # BatchNorm, then the ReLU of :113
            else:
                x = mod(x, eng)
        l0 = self.line0_conv
        enc_x[-1] = eng.conv2d(enc_x[-1], l0) + l0.bias.to(x.dtype).view(1, -1, 1, 1)
        for dec_block, ex in zip(list(self.decoder_blocks)[::-1], enc_x[::-1]):
            x = dec_block(x, ex, eng)
        return unet_tail(self, x, eng)

    def forward(self, x):
        return self.forward_lowres(x)

    def pretrained_parameters(self):
        if self.pretrained:
            return list(self.base_model.features.parameters())
        return []

    def new_parameters(self):
        if self.pretrained:
            pretrained_ids = [id(p) for p in self.base_model.features.parameters()]
            return [p for p in self.parameters() if id(p) not in pretrained_ids]
        return list(self.parameters())

    def freeze_batchnorm(self):
        self.base_model.apply(freeze_bn_module)


def densenet161unet(num_classes):
    return DenseUNet(tv_backbones.densenet161(), num_classes, mean=None, std=None, pretrained=False)


def densenet161unet_imagenet(num_classes, pretrained=True):
    """The reference's factory always downloads the ImageNet DenseNet-161 (denseunet.py:150-153); `pretrained=False`
    (an addition) builds the same module tree with `pretrained` = True semantics for the parameter groups but random
    encoder weights -- what synthetic runs and tests use."""
    if pretrained:
        raise NotImplementedError('pretrained ImageNet weights for the torchvision DenseNet-161 cannot be downloaded here '
                                  '(no network); pass pretrained=False and load a state dict (keys "base_model.features.*")')
    mean = np.array([0.485, 0.456, 0.406])
    std = np.array([0.229, 0.224, 0.225])
    return DenseUNet(tv_backbones.densenet161(), num_classes, mean=mean, std=std, pretrained=True)


for _cls in (DecoderBlock, DenseUNet):
    _cls.__module__ = 'architectures.denseunet'
