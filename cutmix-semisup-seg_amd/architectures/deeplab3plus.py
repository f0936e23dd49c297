"""
DeepLab v3+ (ResNet-101, output stride 8) with the reference's module tree and state-dict keys
(architectures/deeplab3plus.py:26-164).

The reference assembles this model from torchvision 0.5.0 parts (`resnet.resnet101(replace_stride_with_dilation=
[False, True, True])`, `IntermediateLayerGetter`, `segmentation.deeplabv3.ASPP`, deeplab3plus.py:13-15,89-98);
torchvision is not part of this build, so those parts are restated here from their published structure (parity is
UNPINNED for this row -- no vectors exist on the reference side; the independent CPU restatement in
oracle/deeplab3plus.py is the checker):

  backbone   ResNet v1.5 bottlenecks (stride / dilation on the 3x3), layer3 / layer4 dilated instead of strided
             (first block of a dilated layer keeps the previous dilation), stem max-pool without ceil_mode;
             taps: layer1 -> 'low_level' (256 ch, 1/4), layer4 -> 'out' (2048 ch, 1/8)
  head       DeepLabHeadV3Plus (deeplab3plus.py:26-64): project 256->48, ASPP(2048, [12, 24, 36]) incl. the global
             pooling branch and Dropout(0.5), two 3x3 conv-BN-ReLU, 1x1 -> classes; Kaiming-normal init of every
             head convolution (:58-64)
  wrapper    DeepLabv3Wrapper (:104-158): BLOCK_SIZE / MEAN / STD, `freeze_batchnorm()` freezes the BACKBONE only
             (:120-121), `pretraining=None` => pretrained_parameters() == [] and every parameter trains at the full
             learning rate (:138-151, factory :162-164)

Execution: bf16 channels-last through the layer engine (HipConvEngine below: hand-written kernels); the low-resolution
logits (1/4 of the input) are handed to the fused loss / evaluation kernels, which apply the final bilinear upsample
(align_corners=False, deeplab3plus.py:77) in-kernel. With batch-statistics BatchNorm and dropout in the head the
samples of a batch are not independent, so the training step keeps the reference's separate passes
(step.py: fuse_batches=False). The backbone on the hand-written MFMA convolution kernels is the next step for this
row.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .util import freeze_bn_module
from .deeplab2 import LayerEngine


class Bottleneck(nn.Module):
    """torchvision ResNet v1.5 bottleneck: the 3x3 carries stride and dilation."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=dilation, dilation=dilation,
                               bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x, eng):
        out = eng.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        out = eng.conv_bn_act(out, self.conv2, self.bn2, relu=True)
        res = x if self.downsample is None else eng.conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False)
        return eng.conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=res)


class ResNetTaps(nn.ModuleDict):
    """ResNet-101 up to layer4 in the child order of IntermediateLayerGetter (keys conv1, bn1, relu, maxpool,
    layer1..layer4); returns {'low_level': layer1, 'out': layer4} (deeplab3plus.py:96-98)."""

    def __init__(self, layers=(3, 4, 23, 3)):
        super(ResNetTaps, self).__init__()
        self.inplanes, self.dilation = 64, 1
        self['conv1'] = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self['bn1'] = nn.BatchNorm2d(64)
        self['relu'] = nn.ReLU(inplace=True)
        self['maxpool'] = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self['layer1'] = self._make_layer(64, layers[0])
        self['layer2'] = self._make_layer(128, layers[1], stride=2)
        self['layer3'] = self._make_layer(256, layers[2], stride=2, dilate=True)
        self['layer4'] = self._make_layer(512, layers[3], stride=2, dilate=True)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        stages = [Bottleneck(self.inplanes, planes, stride, previous_dilation, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            stages.append(Bottleneck(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*stages)

    def forward(self, x, eng):
        x = eng.conv_bn_act(x, self['conv1'], self['bn1'], relu=True)
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        out = {}
        for name, tap in (('layer1', 'low_level'), ('layer2', None), ('layer3', None), ('layer4', 'out')):
            for blk in self[name]:
                x = blk(x, eng)
            if tap is not None:
                out[tap] = x
        return out


class ASPP(nn.Module):
    """torchvision.models.segmentation.deeplabv3.ASPP: keys convs.{0..3}.{0,1}, convs.4.{1,2}, project.{0,1}."""

    def __init__(self, in_channels, atrous_rates, out_channels=256):
        super(ASPP, self).__init__()
        mods = [nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, bias=False), nn.BatchNorm2d(out_channels),
                              nn.ReLU())]
        for r in atrous_rates:
            mods.append(nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, padding=r, dilation=r, bias=False),
                                      nn.BatchNorm2d(out_channels), nn.ReLU()))
        mods.append(nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(in_channels, out_channels, 1, bias=False),
                                  nn.BatchNorm2d(out_channels), nn.ReLU()))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(nn.Conv2d(len(mods) * out_channels, out_channels, 1, bias=False),
                                     nn.BatchNorm2d(out_channels), nn.ReLU(), nn.Dropout(0.5))

    def _branches(self, xs, convs, eng, on_device):
        """The 1 x 1 and the three dilated 3 x 3 branches. (round 6) The branches are independent until the concat, and at 33 x 33 a
        dilated 3 x 3 over 2048 channels is ONE 85-tile launch on 256 CUs (0.9 ms each, forward; 0.65 ms its data gradient): two of
        the three heavy branches run on the pooled weight-gradient streams (idle in a forward pass), forked from the current
        stream and joined before the concat. autograd runs a node's backward on the stream its forward ran on, so the branches'
        backward chains overlap the same way. CMS_ASPP_STREAMS=0: one after the other on the current stream (rounds 1-5)."""
        import os
        if not on_device or os.environ.get('CMS_ASPP_STREAMS', '1') == '0' or ops._REC is not None or not ops.side_streams_enabled() or len(convs) < 3:
            return [eng.conv_bn_act(xi, m[0], m[1], relu=True) for xi, m in zip(xs, convs)]
        cur = torch.cuda.current_stream()
        sides = [ops.pooled_stream(xs[0].device, 'wgrad0'), ops.pooled_stream(xs[0].device, 'wgrad1')]
        sides = [s for s in sides if s.cuda_stream != cur.cuda_stream]
        out = [None] * len(convs)
        for i, (xi, m) in enumerate(zip(xs, convs)):
            st = sides[(i - 2) % len(sides)] if (i >= 2 and sides) else None      # branches 0 (1 x 1) and 1 stay on the current stream
            if st is None:
                out[i] = eng.conv_bn_act(xi, m[0], m[1], relu=True)
                continue
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                out[i] = eng.conv_bn_act(xi, m[0], m[1], relu=True)
            xi.record_stream(st)
            out[i].record_stream(cur)
        for st in sides:
            cur.wait_stream(st)
        return out

    def forward(self, x, eng):
        convs = list(self.convs)
        xh = _as_nhwc(x)
        if xh is not None:
            # hand-written data movement (csrc/nhwc.hip): the five consumers' gradients are summed by one launch, the pooled
            # branch is a row reduction, the concat writes channel slices (the pooled branch broadcast in the same pass)
            xs = [a.permute(0, 3, 1, 2) for a in ops.fanout(xh, len(convs))]
        else:
            xs = [x] * len(convs)
        br = self._branches(xs, convs[:-1], eng, xh is not None)
        pool = convs[-1]
        if xh is not None:
            g = ops.global_avg_pool(xs[-1].permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        else:
            g = x.float().mean(dim=(2, 3), keepdim=True).to(x.dtype)
        g = eng.conv_bn_act(g, pool[1], pool[2], relu=True)
        brh = [_as_nhwc(b) for b in br]
        if xh is not None and all(b is not None for b in brh) and g.is_cuda and g.shape[1] % 8 == 0 and g.dtype == br[0].dtype:
            cat = ops.concat_channels(brh + [g.permute(0, 2, 3, 1)]).permute(0, 3, 1, 2)
        else:
            # bilinear upsampling of a 1 x 1 map is a broadcast (torchvision's F.interpolate call computes the same values);
            # the library's channels-last bilinear BACKWARD funnels H*W atomics into one pixel here: 190 ms per call
            cat = torch.cat(br + [g.expand(-1, -1, x.shape[2], x.shape[3])], dim=1)
        y = eng.conv_bn_act(cat, self.project[0], self.project[1], relu=True)
        drop = self.project[3]
        return F.dropout(y, drop.p, drop.training)


def _as_nhwc(t):
    """NHWC view of a channels-last (N,C,H,W) CUDA tensor that csrc/nhwc.hip can take, else None."""
    if not (t.is_cuda and t.dim() == 4 and t.dtype in (torch.bfloat16, torch.float32) and t.shape[1] % 8 == 0):
        return None
    v = t.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else None


class DeepLabHeadV3Plus(nn.Module):
    def __init__(self, in_channels, low_level_channels, num_classes, aspp_dilate=(12, 24, 36)):
        super(DeepLabHeadV3Plus, self).__init__()
        self.project = nn.Sequential(nn.Conv2d(low_level_channels, 48, 1, bias=False), nn.BatchNorm2d(48),
                                     nn.ReLU(inplace=True))
        self.aspp = ASPP(in_channels, list(aspp_dilate))
        self.classifier = nn.Sequential(
            nn.Conv2d(304, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
            nn.Conv2d(256, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
            nn.Conv2d(256, num_classes, 1))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, feature, eng):
        low = eng.conv_bn_act(feature['low_level'], self.project[0], self.project[1], relu=True)
        out = self.aspp(feature['out'], eng)
        lh, oh = _as_nhwc(low), _as_nhwc(out)
        if lh is not None and oh is not None and lh.dtype == oh.dtype:
            # upsample written straight into its channel slice of the concat buffer, gather-form adjoint (csrc/nhwc.hip)
            y = ops.upsample_concat(lh, oh, align_corners=False).permute(0, 3, 1, 2)
        else:
            # (NCHW-contiguous detour: the library's channels-last bilinear backward is ~100x slower than its NCHW one)
            out = F.interpolate(out.contiguous(), size=low.shape[2:4], mode='bilinear', align_corners=False)
            y = torch.cat([low, _ToChannelsLast.apply(out)], dim=1)
        y = eng.conv_bn_act(y, self.classifier[0], self.classifier[1], relu=True)
        y = eng.conv_bn_act(y, self.classifier[3], self.classifier[4], relu=True)
        last = self.classifier[6]
        clf = getattr(eng, 'classifier', None)
        if clf is None:
            raise RuntimeError('engine {} has no classifier kernel for {}'.format(type(eng).__name__, last))
        return clf(y, last)                                     # MFMA kernel, fp32 NCHW logits from the epilogue


class DeepLabV3Plus(nn.Module):
    def __init__(self, backbone, classifier):
        super(DeepLabV3Plus, self).__init__()
        self.backbone = backbone
        self.classifier = classifier


class _ToChannelsLast(torch.autograd.Function):
    """NCHW-contiguous -> channels-last, with the gradient handed back NCHW-contiguous: keeps the library's bilinear
    upsample on its NCHW kernels in BOTH directions (its channels-last backward is ~100x slower on gfx950)."""

    @staticmethod
    def forward(ctx, x):
        return x.contiguous(memory_format=torch.channels_last)

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


class HipConvEngine(LayerEngine):
    """The engine of the networks that run layer by layer (DeepLab v3+ head, the U-Nets, batch-statistics passes the executor does
    not take), in bf16 (throughput) or fp32 (parity configuration, csrc/conv_f32.hip). EVERY convolution runs on the hand-written
    kernels (backbone_hip.hip_conv2d: MFMA implicit GEMM; channel padding, tap chunks, strided phases for the layers that need them)
    and every batch-statistics BatchNorm on csrc/bn.hip; a layer neither can express raises -- there is no library fallback in this
    package (round 6).

      strict = False ('auto')  the default engine;
      strict = True  ('hip')   additionally refuses convolutions that are not registered layers of the network and checks the
                               BatchNorm preconditions up front. This is what the oracle comparisons of these networks run on.
    """

    def __init__(self, dtype, wrapper, strict=False):
        super(HipConvEngine, self).__init__(dtype)
        from ..arena import ensure_arena
        self.strict = strict
        self.arena = ensure_arena(wrapper, with_grad=any(p.requires_grad for p in wrapper.parameters()),
                                  with_bf16=(dtype == torch.bfloat16))
        self.keys = {id(m): name + '.weight' for name, m in wrapper.named_modules() if isinstance(m, nn.Conv2d)}
        self.library_convs = 0          # convolutions handed to a library: none, by construction (kept for the tests' asserts)

    def prepare_input(self, x):
        return x.to(dtype=self.dtype, memory_format=torch.channels_last)

    def conv2d(self, x, conv):
        from ..backbone_hip import hip_conv2d, hip_conv2d_eligible
        key = self.keys.get(id(conv))
        if key is None:
            raise RuntimeError('convolution {} is not a registered layer of this network (no weight slice in its arena)'.format(conv))
        if not self.strict and not hip_conv2d_eligible(x, conv, self.dtype):
            raise RuntimeError('convolution {} on input {} cannot be expressed by the hand-written kernels (ungrouped, square kernel, '
                               'symmetric stride / padding / dilation needed); there is no library fallback'.format(key, tuple(x.shape)))
        return hip_conv2d(x, conv, self.arena, key, self.dtype)

    def bn_act(self, y, bn, relu, residual=None):
        if self.strict and bn is not None and bn.training:
            yh = y.permute(0, 2, 3, 1)
            if not (yh.is_contiguous() and y.shape[1] % 8 == 0 and bn.momentum is not None and bn.running_mean is not None):
                raise RuntimeError('engine_kind = "hip": BatchNorm over {} channels ({}) has no hand-written kernel '
                                   '(channels-last input with channels % 8 == 0 needed)'.format(y.shape[1], bn))
        if bn is not None and bn.training:
            self.__dict__.get('_frozen_affines', {}).pop(id(bn), None)      # (its running statistics move: a cached affine is stale)
        if (bn is not None and not bn.training and y.is_cuda and y.shape[1] % 8 == 0 and not bn.weight.requires_grad
                and not bn.bias.requires_grad and bn.running_mean is not None and y.dtype in (torch.bfloat16, torch.float32)
                and os.environ.get('CMS_FROZEN_BN_FUSED', '1') != '0'):
            # (round 6) eval-mode BatchNorm with a non-trainable affine (the teacher of the VAT trainer: three passes over ~170 such
            # layers per iteration of the DenseNet-161 U-Net) as ONE launch: the tensor-op form below is ~9 launches forward (rsqrt,
            # two multiplies, subtract, two casts, addcmul, add, ReLU) and 4 backward. scale / shift: once per weight version.
            yh = y.permute(0, 2, 3, 1)
            if yh.is_contiguous():
                rh = None
                if residual is not None:
                    rh = residual.permute(0, 2, 3, 1)
                    rh = rh if rh.is_contiguous() else rh.contiguous()
                scale, shift = self._frozen_affine(bn)
                return ops.frozen_bn_act(yh, scale, shift, relu=relu, res=rh).permute(0, 3, 1, 2)
        return super(HipConvEngine, self).bn_act(y, bn, relu, residual)

    def _frozen_affine(self, bn):
        """fp32 (scale, shift) of an eval-mode BatchNorm, cached per weight version of the network's arena (optimizer / EMA steps
        `touch()` it; a capture into a hipGraph marks it stale first, so the refresh is part of the graph)."""
        cache = self.__dict__.setdefault('_frozen_affines', {})
        hit = cache.get(id(bn))
        if hit is not None and hit[0] == self.arena.version:
            return hit[1], hit[2]
        with torch.no_grad():
            scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
            shift = bn.bias.float() - bn.running_mean.float() * scale
        cache[id(bn)] = (self.arena.version, scale, shift)
        return scale, shift

    def aspp_head(self, x, convs):
        """DeepLab v2's head (deeplab2.py:124-128: the live dilated 3x3 branches, class axis padded to 64, summed; biases
        added in fp32) on the hand-written kernels."""
        from ..backbone_hip import hip_conv2d
        out = None
        for conv in convs:
            y = hip_conv2d(x, conv, self.arena, self.keys[id(conv)], self.dtype).float()
            out = y if out is None else out + y
        bias = sum(c.bias for c in convs)
        return out + bias.view(1, -1, 1, 1)

    def classifier(self, x, conv):
        """1x1 convolution with bias to <= 64 classes -> fp32 NCHW logits (convolution epilogue)."""
        from ..backbone_hip import hip_classifier
        if not (x.is_cuda and x.dtype == self.dtype and conv.out_channels <= 64 and conv.in_channels % 64 == 0):
            raise RuntimeError('classifier {} has no hand-written kernel (<= 64 classes, input channels % 64 == 0, {} input '
                               'needed)'.format(conv, self.dtype))
        key = self.keys[id(conv)]
        return hip_classifier(x, conv, self.arena, key, key[:-len('weight')] + 'bias', self.dtype)


def _engine_of(net, x, strict=None):
    """Engine selection shared by DeepLabv3Wrapper and the U-Nets (`EngineNetMixin`): an explicit `net.engine` object (how the
    tests plug in their library comparison engine), otherwise the HipConvEngine of (compute dtype, strictness). `strict=True`
    asks for the strict engine whatever `engine_kind` says."""
    if net.engine is not None and strict is None:
        return net.engine
    if not x.is_cuda:
        raise RuntimeError('cutmix-semisup-seg_amd networks run on the GPU only (input on {}); there is no CPU '
                           'fallback'.format(x.device))
    if net.compute_dtype not in (torch.bfloat16, torch.float32):
        raise TypeError('compute_dtype must be torch.bfloat16 or torch.float32')
    if net.engine_kind == 'torch':
        raise RuntimeError("engine_kind 'torch' (library convolutions) is not part of this package any more: the comparison engine "
                           "lives in tests/_library_engine.py -- `net.engine = LibraryEngine(dtype)`")
    strict = (net.engine_kind == 'hip') if strict is None else bool(strict)
    # ('auto' in fp32 is the hand-written fp32 engine too -- csrc/conv_f32.hip)
    engines = net.__dict__.setdefault('_hip_engines', {})
    ek = (net.compute_dtype, strict)
    eng = engines.get(ek)
    if eng is None:
        eng = engines[ek] = HipConvEngine(net.compute_dtype, net, strict=strict)
        if not net.__dict__.get('_hip_engine_hooked', False):
            # weights loaded behind the arena's back: refresh the bf16 operand copy
            net.register_load_state_dict_post_hook(lambda module, incompatible: module._cms_arena.refresh_bf16())
            net.__dict__['_hip_engine_hooked'] = True
    net._hip_engine = eng
    return eng


class EngineNetMixin(object):
    """Execution plumbing shared by the networks that run layer by layer through an engine object (the U-Nets):
    compute dtype, engine selection, runtime attributes that survive whole-module pickles (checkpoint.py)."""

    def _init_runtime(self):
        d = self.__dict__
        d.setdefault('compute_dtype', torch.bfloat16)
        d.setdefault('engine', None)            # set to an engine object to override the default
        # 'auto' / 'hip': hand-written kernels for every convolution and BatchNorm ('hip' = strict checks); an explicit
        # `engine` object overrides (the tests' library comparison engine)
        d.setdefault('engine_kind', 'auto')
        d.setdefault('_hip_engine', None)
        d.setdefault('_hip_engines', {})

    def __setstate__(self, state):
        super(EngineNetMixin, self).__setstate__(state)
        self._init_runtime()

    def _engine(self, x):
        return _engine_of(self, x)


class DeepLabv3Wrapper(nn.Module):
    BLOCK_SIZE = (1, 1)
    MEAN = np.array([0.485, 0.456, 0.406])
    STD = np.array([0.229, 0.224, 0.225])
    upsample_align_corners = False

    def __init__(self, model, pretraining=None):
        super(DeepLabv3Wrapper, self).__init__()
        self.deeplab = model
        self.pretraining = pretraining
        self._init_runtime()

    def _init_runtime(self):
        """Execution state of this build (not part of the reference's module): set at construction and again after
        unpickling a whole-module checkpoint, which does not carry it (checkpoint.py strips it on export)."""
        d = self.__dict__
        d.setdefault('compute_dtype', torch.bfloat16)
        d.setdefault('engine', None)
        # 'auto' / 'hip': hand-written kernels for every convolution and BatchNorm ('hip' = strict checks);
        # 'hip_nograd': backbone executor only for passes without gradients; an explicit `engine` object overrides
        d.setdefault('engine_kind', 'auto')
        d.setdefault('_hip_executor', None)
        d.setdefault('_hip_executors', {})
        d.setdefault('_hip_engine', None)
        d.setdefault('_hip_engines', {})

    def __setstate__(self, state):
        super(DeepLabv3Wrapper, self).__setstate__(state)
        self._init_runtime()

    # ------------------------------------------------------------------------------------------ execution
    def _engine(self, x):
        return _engine_of(self, x)

    def _use_hip_backbone(self):
        """The MFMA executor (backbone_hip.DeepLabV3PlusBackboneExecutor) runs the backbone whenever its BatchNorm
        statistics are frozen -- training passes included -- in bf16 (throughput) or, with engine_kind 'hip', in fp32
        (parity configuration); `engine_kind = 'hip_nograd'` restricts it to passes that need no gradient, an explicit
        `engine` object switches it off. With engine_kind 'hip' a backbone on batch statistics runs layer by layer on the strict engine."""
        if self.engine is not None:
            return False
        if self.compute_dtype == torch.float32 and self.engine_kind not in ('hip', 'auto'):
            return False
        if self.engine_kind == 'hip_nograd' and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return False
        return all(not m.training for m in self.deeplab.backbone.modules() if isinstance(m, nn.BatchNorm2d))

    def hip_executor(self):
        ex = self._hip_executors.get(self.compute_dtype)
        if ex is None:
            from ..backbone_hip import DeepLabV3PlusBackboneExecutor
            ex = self._hip_executors[self.compute_dtype] = DeepLabV3PlusBackboneExecutor(self, dtype=self.compute_dtype)
        self._hip_executor = ex
        return ex

    def forward_lowres(self, x):
        """(N,3,H,W) -> fp32 (N,C,h,w) logits at the low-level feature size (the reference's tensor just before its
        final interpolate, deeplab3plus.py:76)."""
        eng = self._engine(x)
        eng.bn_groups = self.sample_groups()        # batch-statistics layers normalise each sample group apart (step.py)
        try:
            if x.is_cuda and self._use_hip_backbone():
                from ..backbone_hip import run_v3_body
                ex = self.hip_executor()
                low, out = run_v3_body(ex, ex.stem(x))          # stem on csrc/stem.hip (floor-mode pool, trainable BN affine)
                feats = {'low_level': low.permute(0, 3, 1, 2), 'out': out.permute(0, 3, 1, 2)}    # channels-last views
            else:
                feats = self.deeplab.backbone(eng.prepare_input(x), eng)
            return self.deeplab.classifier(feats, eng)
        finally:
            eng.bn_groups = 1

    def sample_groups(self):
        """Equal runs of samples the next pass normalises separately in its batch-statistics layers (see
        architectures/deeplab2.py:set_sample_groups); 1 otherwise."""
        return int(self.__dict__.get('_bn_groups', 1))

    def set_sample_groups(self, groups):
        self.__dict__['_bn_groups'] = int(groups)

    def supports_sample_groups(self):
        """True when every BatchNorm that runs on batch statistics (the head always: deeplab3plus.py:120-121; the backbone
        too without --freeze_bn) takes csrc/bn.hip, whose kernels keep sample groups apart: the training step may then send
        [supervised; mixed] through the network as one batch. (Dropout draws per element and couples nothing.)"""
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d) and m.training]
        return (len(bns) > 0 and self.engine is None
                and all(m.num_features % 8 == 0 and m.momentum is not None and m.running_mean is not None for m in bns))

    def forward(self, x, feature_maps=False, use_dropout=False):
        lo = self.forward_lowres(x)
        if lo.is_cuda:
            return ops.upsample_bilinear(lo, x.shape[2:4], align_corners=False)
        return F.interpolate(lo, size=x.shape[2:4], mode='bilinear', align_corners=False)

    def samples_are_independent(self):
        """True when no layer couples the samples of a batch (all BatchNorms frozen, dropout inactive): only then may
        the training step concatenate batches (step.py). Never the case in the reference's training configuration:
        the head's BatchNorms use batch statistics (freeze_batchnorm() covers the backbone only)."""
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d) and m.training:
                return False
            if isinstance(m, nn.Dropout) and m.training and m.p > 0:
                return False
        return True

    # ------------------------------------------------------------------------------------------ reference API
    def freeze_batchnorm(self):
        self.deeplab.backbone.apply(freeze_bn_module)

    def _backbone_parameters(self):
        return list(self.deeplab.backbone.parameters())

    def _classifier_end_parameters(self):
        return list(self.deeplab.classifier.classifier[-1].parameters())

    def pretrained_parameters(self):
        if self.pretraining is None:
            return []
        elif self.pretraining == 'imagenet':
            return self._backbone_parameters()
        elif self.pretraining == 'coco':
            new_ids = [id(p) for p in self._classifier_end_parameters()]
            return [p for p in self.parameters() if id(p) not in new_ids]
        else:
            raise ValueError('Unknown pretraining {}'.format(self.pretraining))

    def new_parameters(self):
        if self.pretraining is None:
            return list(self.parameters())
        elif self.pretraining == 'imagenet':
            backbone_ids = [id(p) for p in self._backbone_parameters()]
            return [p for p in self.parameters() if id(p) not in backbone_ids]
        elif self.pretraining == 'coco':
            return self._classifier_end_parameters()
        else:
            raise ValueError('Unknown pretraining {}'.format(self.pretraining))


def _deeplabv3plus(num_classes, output_stride=8, layers=(3, 4, 23, 3)):
    if output_stride != 8:
        raise NotImplementedError('only output stride 8 is used by the reference (deeplab3plus.py:163)')
    backbone = ResNetTaps(layers)
    classifier = DeepLabHeadV3Plus(2048, 256, num_classes, (12, 24, 36))
    return DeepLabV3Plus(backbone, classifier)


# whole-module pickles name the reference's OWN classes by the reference's module path (architectures/deeplab3plus.py:26-158;
# the root-level `architectures/deeplab3plus.py` re-exports these very objects). The torchvision parts the reference
# assembles the model from (ResNet, IntermediateLayerGetter, ASPP) are restated in this file and pickle under this
# package's path: torchvision is absent, so the reference's code can take a v3+ checkpoint of this build as a state dict
# (identical keys), not as a whole-module pickle.
for _cls in (DeepLabHeadV3Plus, DeepLabV3Plus, DeepLabv3Wrapper):
    _cls.__module__ = 'architectures.deeplab3plus'


def resnet101_deeplabv3plus_imagenet(num_classes, pretrained=True):
    """deeplab3plus.py:162-164. `pretrained=True` would download torchvision's ImageNet ResNet-101 (no network in this
    build environment): refuse instead of silently training from scratch."""
    if pretrained:
        raise NotImplementedError('pretrained ImageNet weights for the torchvision ResNet-101 cannot be downloaded '
                                  'here; build with pretrained=False and load a state dict (keys '
                                  '"deeplab.backbone.*")')
    return DeepLabv3Wrapper(_deeplabv3plus(num_classes, 8))
