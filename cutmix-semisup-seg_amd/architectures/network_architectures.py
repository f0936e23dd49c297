"""
Mirror of the reference's architectures/network_architectures.py: the `seg` architecture registry
(network_architectures.py:15-41), the factory names registered on it (:44-112), `robust_binary_crossentropy`
(:115-118) and `sigmoid_rampup` (:122-130).

Factories whose backbones are outside the CutMix mean-teacher hot path (plain DeepLab v3 from torchvision, PSPNet from
mit_semseg) stay registered under the reference's names and raise NotImplementedError when called, which is
what the reference itself does when their dependencies are missing (:77-79, mit_csail_semseg.py:24-25).
"""
import sys

import numpy as np
import torch

from . import deeplab2
from . import deeplab3plus
from . import resunet
from . import denseunet


class ArchRegistry(object):
    def __init__(self):
        self.archs = {}

    def register(self, name):
        """
        Usage:

        @registry.register('my_arch')
        def my_arch(...):
            ...
        """
        def deco(arch):
            self.archs[name] = arch
            return arch
        return deco

    def get(self, name):
        return self.archs[name]

    def names(self):
        return self.archs.keys()


seg = ArchRegistry()


def _outside_hot_path(name, needs):
    def factory(num_classes=21, pretrained=True):
        raise NotImplementedError('{} is outside the MI355X hot path of this build (needs {}); see DESIGN.md, '
                                  '"out of scope"'.format(name, needs))
    factory.__name__ = name
    return factory


@seg.register('resnet50unet_imagenet')
def resnet50unet_imagenet(num_classes, pretrained=True):
    return resunet.resnet50unet(num_classes, pretrained=pretrained)


@seg.register('resnet101unet_imagenet')
def resnet101unet_imagenet(num_classes, pretrained=True):
    return resunet.resnet101unet(num_classes, pretrained=pretrained)


@seg.register('densenet161unet')
def densenet161unet(num_classes, pretrained=False):
    # (the reference's factory takes no `pretrained`; accepted and ignored so that the trainers can pass it uniformly)
    return denseunet.densenet161unet(num_classes)


@seg.register('densenet161unet_imagenet')
def densenet161unet_imagenet(num_classes, pretrained=True):
    return denseunet.densenet161unet_imagenet(num_classes, pretrained=pretrained)


for _name, _needs in (('resnet101_deeplabv3_coco', 'torchvision DeepLab v3'),
                      ('resnet101_deeplabv3_imagenet', 'torchvision DeepLab v3'),
                      ('resnet101_pspnet_imagenet', 'the mit_semseg package')):
    seg.register(_name)(_outside_hot_path(_name, _needs))


@seg.register('resnet101_deeplab_coco')
def resnet101_deeplab_coco(num_classes=21, pretrained=True):
    return deeplab2.resnet101_deeplab_coco(num_classes=num_classes, pretrained=pretrained)


@seg.register('resnet101_deeplab_imagenet')
def resnet101_deeplab_imagenet(num_classes=21, pretrained=True):
    return deeplab2.resnet101_deeplab_imagenet(num_classes=num_classes, pretrained=pretrained)


@seg.register('resnet101_deeplab_imagenet_mittal_std')
def resnet101_deeplab_imagenet_mittal_std(num_classes=21, pretrained=True):
    return deeplab2.resnet101_deeplab_imagenet_mittal_std(num_classes=num_classes, pretrained=pretrained)


@seg.register('resnet101_deeplabv3plus_imagenet')
def resnet101_deeplabv3plus_imagenet(num_classes=21, pretrained=True):
    return deeplab3plus.resnet101_deeplabv3plus_imagenet(num_classes=num_classes, pretrained=pretrained)


def robust_binary_crossentropy(pred, tgt, eps=1e-6):
    """-(t*log(p+eps) + (1-t)*log(1-p+eps)), element-wise (tensor utility; the training step uses the fused
    'bce' mode of the consistency kernel instead)."""
    return -(tgt * torch.log(pred + eps) + (1.0 - tgt) * torch.log(1.0 - pred + eps))


EPS = sys.float_info.epsilon


def sigmoid_rampup(current, rampup_length):
    """Exponential rampup from https://arxiv.org/abs/1610.02242: exp(-5 (1 - t/T)^2), 1.0 when T == 0."""
    if rampup_length == 0:
        return 1.0
    t = np.clip(current, 0.0, rampup_length)
    phase = 1.0 - t / rampup_length
    return float(np.exp(-5.0 * phase * phase))
