"""
Encoders of the reference's U-Nets, restated from the published structure of torchvision 0.5.0 (`models.resnet50`,
`models.resnet101`, `models.densenet161`; call sites architectures/resunet.py:112-116, architectures/denseunet.py:147-153):
same module tree, same state_dict keys (incl. the unused ImageNet classifier `fc` / `classifier`), so that torchvision
checkpoints load by name. torchvision itself is not part of this build -- PARITY UNPINNED for these rows;
oracle/unets.py is the independent CPU restatement the tests check against.

Execution goes through an engine object (architectures/deeplab2.py: LayerEngine -> deeplab3plus.py: HipConvEngine):
BatchNorm on batch statistics runs on csrc/bn.hip (+ ReLU fused, SyncBN under torch.distributed), convolutions that fit
the MFMA kernels (stride 1, 'same' padding, >= 128 input channels, output channels in multiples of 64) on csrc/conv.hip,
the rest on the library.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    """ResNet v1.5 bottleneck (the 3x3 carries the stride)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, eng):
        out = eng.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        out = eng.conv_bn_act(out, self.conv2, self.bn2, relu=True)
        res = x if self.downsample is None else eng.conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False)
        return eng.conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=res)


class ResNet(nn.Module):
    """torchvision ResNet (bottleneck variants): conv1, bn1, relu, maxpool, layer1..4, avgpool, fc."""

    def __init__(self, layers, num_classes=1000):
        super(ResNet, self).__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * 4, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        stages = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            stages.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*stages)


def resnet50():
    return ResNet([3, 4, 6, 3])


def resnet101():
    return ResNet([3, 4, 23, 3])


class _DenseLayer(nn.Module):
    """norm1 -> relu1 -> conv1 (1x1, bn_size * growth) -> norm2 -> relu2 -> conv2 (3x3, growth) on the concatenation of
    all earlier feature maps of the block."""

    def __init__(self, num_input_features, growth_rate, bn_size):
        super(_DenseLayer, self).__init__()
        self.norm1 = nn.BatchNorm2d(num_input_features)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(num_input_features, bn_size * growth_rate, kernel_size=1, stride=1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth_rate)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth_rate, growth_rate, kernel_size=3, stride=1, padding=1, bias=False)

    def forward(self, feats, eng):
        x = torch.cat(feats, 1) if len(feats) > 1 else feats[0]
        y = eng.conv2d(eng.bn_act(x, self.norm1, relu=True), self.conv1)
        return eng.conv2d(eng.bn_act(y, self.norm2, relu=True), self.conv2)


class _DenseBlock(nn.ModuleDict):
    def __init__(self, num_layers, num_input_features, bn_size, growth_rate):
        super(_DenseBlock, self).__init__()
        for i in range(num_layers):
            self['denselayer{}'.format(i + 1)] = _DenseLayer(num_input_features + i * growth_rate, growth_rate, bn_size)

    def forward(self, x, eng):
        feats = [x]
        for layer in self.values():
            feats.append(layer(feats, eng))
        return torch.cat(feats, 1)


class _Transition(nn.Sequential):
    def __init__(self, num_input_features, num_output_features):
        super(_Transition, self).__init__()
        self.add_module('norm', nn.BatchNorm2d(num_input_features))
        self.add_module('relu', nn.ReLU(inplace=True))
        self.add_module('conv', nn.Conv2d(num_input_features, num_output_features, kernel_size=1, stride=1, bias=False))
        self.add_module('pool', nn.AvgPool2d(kernel_size=2, stride=2))

    def forward(self, x, eng):
        return F.avg_pool2d(eng.conv2d(eng.bn_act(x, self.norm, relu=True), self.conv), 2, 2)


class DenseNet(nn.Module):
    """torchvision DenseNet: features.{conv0, norm0, relu0, pool0, denseblockK, transitionK, norm5}, classifier."""

    def __init__(self, growth_rate=48, block_config=(6, 12, 36, 24), num_init_features=96, bn_size=4, num_classes=1000):
        super(DenseNet, self).__init__()
        feats = nn.Sequential()
        feats.add_module('conv0', nn.Conv2d(3, num_init_features, kernel_size=7, stride=2, padding=3, bias=False))
        feats.add_module('norm0', nn.BatchNorm2d(num_init_features))
        feats.add_module('relu0', nn.ReLU(inplace=True))
        feats.add_module('pool0', nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
        n = num_init_features
        for i, nl in enumerate(block_config):
            feats.add_module('denseblock{}'.format(i + 1), _DenseBlock(nl, n, bn_size, growth_rate))
            n += nl * growth_rate
            if i != len(block_config) - 1:
                feats.add_module('transition{}'.format(i + 1), _Transition(n, n // 2))
                n //= 2
        feats.add_module('norm5', nn.BatchNorm2d(n))
        self.features = feats
        self.classifier = nn.Linear(n, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.constant_(m.bias, 0)


def densenet161():
    return DenseNet(48, (6, 12, 36, 24), 96)
