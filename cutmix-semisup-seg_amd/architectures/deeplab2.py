"""
Mirror of the reference's architectures/deeplab2.py: DeepLab v2 on ResNet-101 (the network of BASELINE configs 2
and 3), same constructor / attributes / state_dict keys, executed MI355X-first.

Kept from the reference (file:line relative to upstream):
  * structure: 7x7/2 stem, max-pool 3x3/2 ceil_mode (deeplab2.py:140-146); bottlenecks with the stride on the FIRST
    1x1 (Caffe style, :70) and a dilated 3x3 (:76-77); layers [3,4,23,3] with layer3 dilation 2 and layer4 dilation 4
    at stride 1 (:147-150); a downsample branch on every first block (:163-168)
  * ASPP head of four 3x3 convs 2048->C with dilations 6/12/18/24 of which only the first two are ever summed
    because of the early `return` in the reference's loop (:124-128)  [SURVEY Appendix A, Q1]
  * every BatchNorm affine parameter has requires_grad=False (:72-84, 143-144, 170-171); `freeze_batchnorm()` puts
    BN layers in eval mode (:244-245)
  * initialisation: conv weights ~ N(0, 0.01), BN gamma 1 / beta 0 (:153-159)
  * `pretrained_parameters()` yields each backbone conv weight once per enclosing module of the walk (3x for block
    convs, 4x for downsample convs, 1x for the stem) and `new_parameters()` yields the ASPP tensors (:208-242)
  * BLOCK_SIZE / MEAN / STD attributes, the three factory functions and `_load_state_into_model` (:248-322)
  * `forward(x)` returns logits at the input resolution through a bilinear upsample with align_corners=True (:204)

MI355X-first:
  * activations are bf16 NHWC (channels-last) between layers, fp32 accumulation, logits leave the head in fp32
  * frozen BatchNorm is an affine epilogue of the convolution that produced its input (scale/shift folded per
    forward from the fp32 BN tensors) instead of a separate pass over the activation
  * `forward_lowres(x)` exposes the (N,C,h,w) head output so that the loss / evaluation kernels can fuse the
    upsample (ops.py) -- the training step never materialises (N,C,H,W) logits
  * weights are read from the bf16 copy the fused optimizer maintains in the parameter arena when present
"""
import numpy as np
import os
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .util import freeze_bn_module

affine_par = True

_RESNET_101_DEEPLAB_COCO_URL = 'http://vllab1.ucmerced.edu/~whung/adv-semi-seg/resnet101COCO-41f33a49.pth'
_RESNET_101_IMAGENET_URL = 'https://download.pytorch.org/models/resnet101-5d3b4d8f.pth'

ASPP_DILATIONS = (6, 12, 18, 24)
ASPP_LIVE = 2          # number of branches the reference's forward actually sums


def _frozen_bn(n):
    bn = nn.BatchNorm2d(n, affine=affine_par)
    for p in bn.parameters():
        p.requires_grad = False
    return bn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False)
        self.bn1 = _frozen_bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=dilation, bias=False,
                               dilation=dilation)
        self.bn2 = _frozen_bn(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = _frozen_bn(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, eng=None):
        eng = eng or _default_engine(x)
        out = eng.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        out = eng.conv_bn_act(out, self.conv2, self.bn2, relu=True)
        if self.downsample is not None:
            residual = eng.conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False)
        else:
            residual = x
        return eng.conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=residual)


class Classifier_Module(nn.Module):
    def __init__(self, dilation_series, padding_series, num_classes):
        super(Classifier_Module, self).__init__()
        self.conv2d_list = nn.ModuleList()
        for dilation, padding in zip(dilation_series, padding_series):
            self.conv2d_list.append(nn.Conv2d(2048, num_classes, kernel_size=3, stride=1, padding=padding,
                                              dilation=dilation, bias=True))
        for m in self.conv2d_list:
            m.weight.data.normal_(0, 0.01)

    def forward(self, x, eng=None):
        eng = eng or _default_engine(x)
        # conv_d6(x) + conv_d12(x): the reference returns from inside its accumulation loop after one iteration
        return eng.aspp_head(x, [self.conv2d_list[i] for i in range(ASPP_LIVE)])


class LayerEngine(object):
    """
    Base of the engines that execute conv(+BatchNorm)(+residual)(+ReLU) units layer by layer in `dtype`, channels-last. It owns what
    every such engine shares -- batch-statistics BatchNorm (+ residual + ReLU) on csrc/bn.hip, frozen statistics as an affine,
    input preparation, the stem's ceil-mode pool -- and leaves the CONVOLUTIONS to the subclass: `deeplab3plus.HipConvEngine` runs
    them on the hand-written kernels. There is no library (MIOpen) convolution engine in this package (round 6): the comparison
    engine the GPU tests A/B against lives in tests/_library_engine.py and is plugged in through `net.engine = ...`. The
    frozen-BatchNorm DeepLab v2 does not come here at all: it runs on the static MFMA executor (backbone_hip.py).
    """

    def __init__(self, dtype=torch.bfloat16):
        self.dtype = dtype

    def _weight(self, conv):
        w = conv.weight
        if w.dtype != self.dtype:
            w = w.to(self.dtype)
        return w

    def prepare_input(self, x):
        return x.to(dtype=self.dtype, memory_format=torch.channels_last)

    def conv2d(self, x, conv):
        raise NotImplementedError('{} executes no convolutions: use deeplab3plus.HipConvEngine (the hand-written kernels)'.format(
            type(self).__name__))

    def aspp_head(self, x, convs):
        raise NotImplementedError('{} has no ASPP head'.format(type(self).__name__))

    def conv_bn_act(self, x, conv, bn, relu, residual=None):
        return self.bn_act(self.conv2d(x, conv), bn, relu, residual)

    def _bn_on_hip(self, y, bn):
        """True when csrc/bn.hip takes this batch-statistics BatchNorm: device tensor, channels-last, channels % 8 == 0, running
        statistics with a momentum."""
        return (y.is_cuda and y.permute(0, 2, 3, 1).is_contiguous() and y.shape[1] % 8 == 0 and bn.momentum is not None
                and bn.running_mean is not None)

    def bn_act(self, y, bn, relu, residual=None):
        """relu(bn(y) (+ residual)): batch statistics (training mode) on csrc/bn.hip, frozen statistics as an affine."""
        if bn is not None:
            if bn.training:
                if not self._bn_on_hip(y, bn):
                    raise RuntimeError('BatchNorm over {} channels ({}) has no hand-written kernel (device tensor, channels-last, '
                                       'channels % 8 == 0, running statistics needed); there is no library fallback'.format(
                                           y.shape[1], bn))
                # batch-statistics BatchNorm (+ residual + ReLU) on csrc/bn.hip; under torch.distributed its statistics are
                # all-reduced (SyncBN)
                yh = y.permute(0, 2, 3, 1)                 # NHWC view of the channels-last tensor
                rh = None
                if residual is not None:
                    rh = residual.permute(0, 2, 3, 1)
                    rh = rh if rh.is_contiguous() else rh.contiguous()
                groups = int(getattr(self, 'bn_groups', 1))    # sample groups normalised apart (step.py, grouped passes)
                out = ops.batch_norm_act(yh, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                         relu=relu, res=rh, groups=groups)
                if bn.num_batches_tracked is not None:
                    bn.num_batches_tracked += groups
                return out.permute(0, 3, 1, 2)
            scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            shift = bn.bias - bn.running_mean * scale
            y = torch.addcmul(shift.to(y.dtype).view(1, -1, 1, 1), y, scale.to(y.dtype).view(1, -1, 1, 1))
        if residual is not None:
            y = y + residual
        if relu:
            y = F.relu(y, inplace=True)
        return y

    def maxpool(self, x):
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1, ceil_mode=True)


def _default_engine(x):
    raise RuntimeError('a Bottleneck / Classifier_Module of this build is executed by its network\'s engine (forward(x, eng)); '
                       'call the network, or pass an engine object')


class ResNetDeepLab(nn.Module):
    BLOCK_SIZE = (1, 1)

    def __init__(self, block, layers, num_classes, mean, std):
        self.MEAN = mean
        self.STD = std
        self.inplanes = 64
        super(ResNetDeepLab, self).__init__()
        self.num_classes = num_classes
        self._init_runtime()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = _frozen_bn(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1, ceil_mode=True)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=1, dilation=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=1, dilation=4)
        self.layer5 = Classifier_Module(list(ASPP_DILATIONS), list(ASPP_DILATIONS), num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, 0.01)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _init_runtime(self):
        """Execution state of this build (not part of the reference's module): set at construction and again after
        unpickling a whole-module checkpoint (reference-made pickles do not carry it, checkpoint.py)."""
        d = self.__dict__
        d.setdefault('compute_dtype', torch.bfloat16)
        d.setdefault('engine', None)          # set to an engine object to override the default executor
        # 'auto' / 'hip': hand-written MFMA executor (backbone_hip.py) for the body (bf16 = throughput configuration, fp32 =
        # parity configuration), the hand-written layer engine for the passes it does not take; 'hip' additionally refuses
        # anything it cannot express. (The library comparison engine of the tests is an `engine` OBJECT: tests/_library_engine.py)
        d.setdefault('engine_kind', 'auto')
        d.setdefault('stem_kind', 'hip')      # 'engine': the stem through the layer engine instead of csrc/stem.hip (comparison runs)
        d.setdefault('_hip_executor', None)   # the executor used last
        d.setdefault('_hip_executors', {})    # compute dtype -> executor
        d.setdefault('_hip_engine', None)     # the layer-by-layer engine used last (batch-statistics passes)
        d.setdefault('_hip_engines', {})
        if 'num_classes' not in d:            # a reference-made pickle: read it off the head
            d['num_classes'] = self.layer5.conv2d_list[0].out_channels

    def __setstate__(self, state):
        super(ResNetDeepLab, self).__setstate__(state)
        self._init_runtime()

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion or dilation == 2 or dilation == 4:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                _frozen_bn(planes * block.expansion))
        stages = [block(self.inplanes, planes, stride, dilation=dilation, downsample=downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            stages.append(block(self.inplanes, planes, dilation=dilation))
        return nn.Sequential(*stages)

    # ------------------------------------------------------------------------------------------ execution
    def _engine(self, x):
        """The layer-by-layer engine of the passes the static executor does not take: BatchNorm on BATCH STATISTICS (the
        reference CLI's default, no --freeze_bn: deeplab2.py:72-84, train_seg_semisup_mask_mt.py:268-275,587). 'auto' in
        bf16: MFMA kernels for the convolutions that fit them well, csrc/bn.hip for BatchNorm; 'hip': every convolution
        (stem as tap chunks, strided 1x1s, the class-wide head) and every BatchNorm on the hand-written kernels, in bf16
        or fp32, or an error. An explicit `self.engine` object overrides both (the tests' library comparison engine)."""
        from .deeplab3plus import _engine_of
        return _engine_of(self, x)

    def _frozen_bn(self):
        # (asked several times per iteration: the list of BatchNorm modules is built once -- the module tree is static --
        # instead of walking ~300 modules each time, which cost 0.4 ms of host time per call at the head of every
        # iteration, where the GPU waits for the host: profiles/r04m_host_profile.txt)
        bns = self.__dict__.get('_bn_list')
        if bns is None:
            bns = self.__dict__['_bn_list'] = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        return all(not m.training for m in bns)

    def _use_hip_body(self):
        """True: the static MFMA executor (backbone_hip.DeepLabHipExecutor) runs the body and the head. With every BatchNorm
        frozen the running statistics fold into the convolution epilogues (and the stem runs on csrc/stem.hip); with
        BatchNorm on batch statistics (round 3) every unit is  conv -> csrc/bn.hip  inside the same recorded programs, the stem
        goes through the layer engine. Under torch.distributed the units all-reduce their per-group sums between the reduction
        and the finalisation (SyncBN: a host op between two launches of the recorded pass); `batchstat_executor = False` forces
        the layer engine."""
        if self.engine is not None:
            return False
        ok = self.compute_dtype in (torch.bfloat16, torch.float32) and self.num_classes <= 32
        if self._frozen_bn():
            if self.engine_kind == 'hip' and not ok:
                raise RuntimeError('the MFMA executor needs bf16 or fp32 compute and <= 32 classes')
            return ok
        if not ok or not self.__dict__.get('batchstat_executor', True):
            return False
        if any(p.requires_grad for m in self.__dict__['_bn_list'] for p in m.parameters(recurse=False)):
            # the executor's batch-statistics backward computes sum(dy) / sum(dy * xhat) for the data gradient only and does not
            # accumulate them into the gradients of a TRAINABLE BatchNorm affine (the reference freezes it: deeplab2.py:76-84);
            # a network with a trainable affine goes through the layer engine, whose autograd function returns dgamma / dbeta
            if self.engine_kind == 'hip':
                raise RuntimeError('batch-statistics passes on the executor need the BatchNorm affine frozen (requires_grad = '
                                   'False, as the reference has it); use engine_kind = "auto" for a trainable affine')
            return False
        return True            # (round 4: under torch.distributed the executor's units all-reduce their statistics -- SyncBN)

    def hip_executor(self):
        # one executor per compute dtype (bf16: throughput configuration; fp32: parity configuration on the f32-input
        # MFMA, csrc/conv_f32.hip -- also what the VAT direction pass switches to inside a bf16 iteration)
        ex = self._hip_executors.get(self.compute_dtype)
        if ex is None:
            from ..backbone_hip import DeepLabHipExecutor
            ex = self._hip_executors[self.compute_dtype] = DeepLabHipExecutor(self, dtype=self.compute_dtype)
        self._hip_executor = ex
        return ex

    def stem_nhwc(self, x):
        """conv1 + bn1 + ReLU + max-pool (:183-186) -> NHWC, the input of the MFMA executor."""
        if self.stem_kind == 'hip' and self._use_hip_body() and self._frozen_bn():
            return self.hip_executor().stem(x)                 # csrc/stem.hip, no library convolution in the pass
        if self.stem_kind == 'hip' and self._use_hip_body() and self.engine_kind == 'auto':
            # batch-statistics pass on the executor: the 7 x 7 / stride 2 stem on the hand-written kernels as well (tap
            # chunks, 3 -> 64 channel padding): the library's weight gradient for this layer takes 36 ms per call at
            # 20 x 321 x 321 (profiles/r03r_*), twice the whole frozen-BatchNorm step
            from .deeplab3plus import _engine_of
            eng = _engine_of(self, x, strict=True)
        else:
            eng = self._engine(x)
        x = eng.prepare_input(x)
        eng.bn_groups = self.sample_groups()
        try:
            x = eng.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        finally:
            eng.bn_groups = 1
        x = eng.maxpool(x)
        return x.permute(0, 2, 3, 1).contiguous()

    def sample_groups(self):
        """Number of equal runs of samples the NEXT batch-statistics pass normalises separately (set by the training step
        around a pass over [supervised batch; mixed batch], see `supports_sample_groups`); 1 otherwise."""
        return int(self.__dict__.get('_bn_groups', 1))

    def set_sample_groups(self, groups):
        self.__dict__['_bn_groups'] = int(groups)

    def supports_sample_groups(self):
        """True when a batch-statistics pass of this network runs entirely on BatchNorm kernels that keep sample groups apart
        (the executor's units and the stem's csrc/bn.hip layer): the training step may then push the reference's separate
        forward passes (train_seg_semisup_mask_mt.py:296-358) through the network as ONE batch."""
        return (not self._frozen_bn()) and self._use_hip_body()

    def forward_lowres(self, x):
        """(N,3,H,W) -> (N,C,h,w) fp32 head output (the reference's `x` just before its interpolate, :193)."""
        if not x.is_cuda:
            raise RuntimeError('cutmix-semisup-seg_amd networks run on the GPU only (input on {}); there is no CPU '
                               'fallback'.format(x.device))
        if self._use_hip_body():
            from ..backbone_hip import run_body
            return run_body(self.hip_executor(), self.stem_nhwc(x))
        eng = self._engine(x)
        x = eng.prepare_input(x)
        x = eng.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        x = eng.maxpool(x)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                x = blk(x, eng)
        return self.layer5(x, eng)

    def forward(self, x, use_dropout=False):
        lo = self.forward_lowres(x)
        return ops.upsample_bilinear(lo, x.shape[2:4], align_corners=True)

    upsample_align_corners = True

    # ------------------------------------------------------------------------------------------ parameter groups
    def pretrained_parameters(self):
        """
        Trainable parameters of everything except the classification head. As in the reference, the walk visits
        every module nested under conv1 / bn1 / layer1..4 and yields each visited module's (recursive) trainable
        parameters, so a tensor comes out once per enclosing module: block conv weights 3x, downsample convs 4x,
        the stem conv once. BatchNorm parameters never appear (requires_grad is False).
        """
        for top in (self.conv1, self.bn1, self.layer1, self.layer2, self.layer3, self.layer4):
            for sub in top.modules():
                for p in sub.parameters():
                    if p.requires_grad:
                        yield p

    def new_parameters(self):
        """The parameters of the classification head (all four ASPP branches)."""
        for p in self.layer5.parameters():
            yield p

    def unused_parameter_keys(self):
        """state_dict keys of parameters that never receive a gradient (ASPP d18 / d24): torch's optimizers skip
        them (grad is None); the fused optimizer needs to be told."""
        keys = []
        for i in range(ASPP_LIVE, len(self.layer5.conv2d_list)):
            keys += ['layer5.conv2d_list.{}.weight'.format(i), 'layer5.conv2d_list.{}.bias'.format(i)]
        return keys

    def freeze_batchnorm(self):
        self.apply(freeze_bn_module)


# whole-module pickles name the classes by module path: the reference's path, so that checkpoints written here load with
# the reference's code and vice versa (the root-level `architectures/deeplab2.py` re-exports these very objects)
for _cls in (Bottleneck, Classifier_Module, ResNetDeepLab):
    _cls.__module__ = 'architectures.deeplab2'


def _hung_mean_std():
    # BGR ImageNet means of the Caffe model flipped to RGB and scaled to [0,1]; std 1/255 re-expands to [0,255]
    mean = np.array((104.00698793, 116.66876762, 122.67891434))[::-1] / 255.0
    std = np.array([1, 1, 1]) / 255.0
    return mean, std


def _load_url(url):
    from torch.utils.model_zoo import load_url
    return load_url(url)


def resnet101_deeplab_coco(num_classes=21, pretrained=True):
    mean, std = _hung_mean_std()
    model = ResNetDeepLab(Bottleneck, [3, 4, 23, 3], num_classes, mean, std)
    if pretrained:
        _load_state_into_model(model, _load_url(_RESNET_101_DEEPLAB_COCO_URL))
    return model


def resnet101_deeplab_imagenet(num_classes=21, pretrained=True):
    mean = np.array([0.485, 0.456, 0.406])
    std = np.array([0.229, 0.224, 0.225])
    model = ResNetDeepLab(Bottleneck, [3, 4, 23, 3], num_classes, mean, std)
    if pretrained:
        _load_state_into_model(model, _load_url(_RESNET_101_IMAGENET_URL))
    return model


def resnet101_deeplab_imagenet_mittal_std(num_classes=21, pretrained=True):
    mean, std = _hung_mean_std()
    model = ResNetDeepLab(Bottleneck, [3, 4, 23, 3], num_classes, mean, std)
    if pretrained:
        _load_state_into_model(model, _load_url(_RESNET_101_IMAGENET_URL))
    return model


def _load_state_into_model(model, state_dict, verbose=False):
    """Copy every tensor of `state_dict` whose name and shape match into the model (others keep their init)."""
    own = model.state_dict()
    with torch.no_grad():
        for name, param in own.items():
            if name not in state_dict:
                if verbose:
                    print('Could not find {}'.format(name))
            elif param.size() != state_dict[name].size():
                if verbose:
                    print('{} -> {}'.format(state_dict[name].shape, param.shape))
            else:
                param.copy_(state_dict[name])
    model.load_state_dict(own)
