"""
Oracle (test infrastructure): DeepLab v2 forward AND backward as an EXPLICIT chain of convolution units, with an
optional model of bf16 STORAGE -- the restatement the bf16 (timed) configuration of the device engine is held to.

Why it exists. `oracle/deeplab2.py` is the fp32 restatement of architectures/deeplab2.py:89-128,183-193 and is pinned by
reference-generated fixtures. The device's throughput configuration keeps every activation, every back-propagated
gradient and the convolution operands in bf16 (fp32 accumulation); ~100 storage roundings of 2^-9 move a loss by 1e-3
..1e-2, so the fp32 oracle cannot tell a wrong tap from storage noise. This module computes THE SAME algorithm with
`storage='bf16'`: values are rounded to bf16 (round-to-nearest-even) exactly where a unit's result is stored --

    forward   a   = R(relu(conv1(x, R(W1)) * s1 + b1))                       deeplab2.py:92-98 (conv, frozen BN, ReLU)
              b   = R(relu(conv2(a, R(W2)) * s2 + b2))                       :99-101
              res = x                    or  R(convd(x, R(Wd)) * sd + bd)    :104-105
              out = R(relu(conv3(b, R(W3)) * s3 + b3 + res))                 :102-103,106-107
              logits = conv_d6(x4, R(W)) + conv_d12(x4, R(W)) + biases       :124-128 (fp32, not rounded)
    backward  (autograd of the above, written out; dC = gradient wrt the pre-ReLU sum of a block, ReLU mask applied)
              dU2 = R(dgrad3(dC, R(W3 * s3)) * [b > 0])      dW3 = s3 * wgrad(b, dC)
              dU1 = R(dgrad2(dU2, R(W2 * s2)) * [a > 0])     dW2 = s2 * wgrad(a, dU2)
              dres = dC  or  R(dgradd(dC, R(Wd * sd)))       dWd = sd * wgrad(x, dC)
              dC' = R((dgrad1(dU1, R(W1 * s1)) + dres) * [x > 0])            dW1 = s1 * wgrad(x, dU1)

with R = identity for `storage='fp32'`. All arithmetic between two roundings is fp32 PyTorch-CPU (ATen convolutions, the
same ops the reference calls); summation order inside a convolution differs from the device's, which is the residual
the GPU tests bound. With R = identity the chain is checked against `oracle/deeplab2.py` + autograd
(tests/test_oracle_chain.py), which ties it to the reference-generated fixtures; the rounding points themselves restate
this repository's storage format, not the reference -- there is nothing upstream to pin them to.

Frozen BatchNorm only (`--freeze_bn`, the configuration of BASELINE configs[1] / [2]).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch.nn.grad import conv2d_input, conv2d_weight

from . import deeplab2 as dl


def _rb(t):
    return t.bfloat16().float()


class Chain(object):
    """`storage`: 'fp32' or 'bf16'. `round_until`: blocks with index >= round_until keep fp32 storage (experiment:
    which tensors feed the error of the head); None = every block rounds."""

    def __init__(self, state, num_classes, layers=dl.LAYERS, storage='bf16', round_until=None):
        if storage not in ('fp32', 'bf16'):
            raise ValueError('storage must be fp32 or bf16')
        self.st = state
        self.C = num_classes
        self.layers = tuple(layers)
        self.storage = storage
        self.round_until = round_until
        self.units = []            # (prefix, stride, dilation, has_down)
        for li, nblk in enumerate(self.layers):
            for b in range(nblk):
                self.units.append(('layer{}.{}'.format(li + 1, b), dl.STRIDES[li] if b == 0 else 1, dl.DILATIONS[li],
                                   b == 0))

    # ------------------------------------------------------------------------------------------ helpers
    def _R(self, t, bi=None):
        if self.storage != 'bf16':
            return t
        if bi is not None and self.round_until is not None and bi >= self.round_until:
            return t
        return _rb(t)

    def _Rw(self, w):
        return _rb(w) if self.storage == 'bf16' else w

    def _affine(self, prefix):
        st = self.st
        scale = st[prefix + '.weight'] * torch.rsqrt(st[prefix + '.running_var'] + dl.BN_EPS)
        bias = st[prefix + '.bias'] - st[prefix + '.running_mean'] * scale
        return scale, bias

    def _stem_weight(self):
        w = self.st['conv1.weight']
        if self.storage != 'bf16':
            return w
        hi = _rb(w)                       # the stem keeps its weights as a bf16 (hi, lo) pair: 16 mantissa bits
        return hi + _rb(w - hi)

    @staticmethod
    def _cba(x, w, scale, bias, stride, dil, k):
        pad = dil * (k - 1) // 2
        y = F.conv2d(x, w, None, stride, pad, dil)
        return y * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)

    # ------------------------------------------------------------------------------------------ units (forward)
    def stem_forward(self, x):
        """-> (s = stored ReLU output of conv1 + bn1, p = max-pooled input of layer1)   deeplab2.py:183-186"""
        sc, bs = self._affine('bn1')
        s = self._R(F.relu(F.conv2d(x, self._stem_weight(), None, 2, 3) * sc.view(1, -1, 1, 1) + bs.view(1, -1, 1, 1)))
        return s, F.max_pool2d(s, 3, 2, 1, ceil_mode=True)

    def block_forward(self, bi, cur, a1_given=None, a2_given=None):
        """One bottleneck on a GIVEN input (deeplab2.py:89-109) -> (a1, a2, out). Feeding the device's own stored input
        here is the teacher-forced per-layer check: only fp32 summation order separates the two results. `a1_given` /
        `a2_given`: the inputs of conv2 / conv3 are forced as well (every convolution checked on its own)."""
        st = self.st
        pre, stride, dil, down = self.units[bi]
        s1, b1 = self._affine(pre + '.bn1')
        s2, b2 = self._affine(pre + '.bn2')
        s3, b3 = self._affine(pre + '.bn3')
        a1 = self._R(F.relu(self._cba(cur, self._Rw(st[pre + '.conv1.weight']), s1, b1, stride, 1, 1)), bi)
        a1_in = a1 if a1_given is None else a1_given
        a2 = self._R(F.relu(self._cba(a1_in, self._Rw(st[pre + '.conv2.weight']), s2, b2, 1, dil, 3)), bi)
        a2_in = a2 if a2_given is None else a2_given
        if down:
            sd, bd = self._affine(pre + '.downsample.1')
            res = self._R(self._cba(cur, self._Rw(st[pre + '.downsample.0.weight']), sd, bd, stride, 1, 1), bi)
        else:
            res = cur
        out = self._R(F.relu(self._cba(a2_in, self._Rw(st[pre + '.conv3.weight']), s3, b3, 1, 1, 1) + res), bi)
        return a1, a2, out

    def head_forward(self, x4):
        """conv_d6 + conv_d12 + biases (deeplab2.py:124-128), fp32 result."""
        st = self.st
        logits = None
        for i, d in enumerate(dl.ASPP_DILATIONS[:2]):
            k = 'layer5.conv2d_list.{}'.format(i)
            y = F.conv2d(x4, self._Rw(st[k + '.weight']), None, 1, d, d)
            logits = y if logits is None else logits + y
        return logits + (st['layer5.conv2d_list.0.bias'] + st['layer5.conv2d_list.1.bias']).view(1, -1, 1, 1)

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x, save=True):
        """x fp32 (N,3,H,W) (already holding bf16-representable values in the bf16 configuration) ->
        (low-resolution logits fp32 (N,C,h,w), saved)."""
        with torch.no_grad():
            s, p = self.stem_forward(x)
            cur = p
            blocks = []
            for bi in range(len(self.units)):
                a1, a2, out = self.block_forward(bi, cur)
                blocks.append((cur, a1, a2))
                cur = out
            logits = self.head_forward(cur)
        saved = dict(x=x, s=s, p_shape=tuple(p.shape), blocks=blocks, x4=cur) if save else None
        return logits, saved

    def block_outputs(self, saved):
        """Input of every bottleneck (= output of the previous one; [0] is the stem output) + the layer4 output."""
        return [b[0] for b in saved['blocks']] + [saved['x4']]

    # ------------------------------------------------------------------------------------------ units (backward)
    def head_backward(self, x4, dlogits, acc):
        """-> dC of the last bottleneck (gradient wrt its pre-ReLU sum, masked); head gradients through `acc`."""
        st = self.st
        nb = len(self.units)
        D = self._R(dlogits)               # the head's backward operand is stored in the activation dtype
        db = dlogits.sum(dim=(0, 2, 3))
        dC = None
        for i, d in enumerate(dl.ASPP_DILATIONS[:2]):
            k = 'layer5.conv2d_list.{}'.format(i)
            w = self._Rw(st[k + '.weight'])
            acc(k + '.weight', conv2d_weight(x4, w.shape, D, 1, d, d))
            acc(k + '.bias', db)
            t = conv2d_input(x4.shape, w, D, 1, d, d)
            dC = t if dC is None else dC + t
        return self._R(dC * (x4 > 0), nb - 1)

    def block_backward(self, bi, xin, a1, a2, dC, acc):
        """Backward of one bottleneck on GIVEN activations and a GIVEN incoming gradient -> dC of the previous one."""
        st = self.st
        pre, stride, dil, down = self.units[bi]
        s1, _ = self._affine(pre + '.bn1')
        s2, _ = self._affine(pre + '.bn2')
        s3, _ = self._affine(pre + '.bn3')
        w1, w2, w3 = (self._Rw(st[pre + '.conv{}.weight'.format(j)]) for j in (1, 2, 3))
        v = lambda s_: s_.view(-1, 1, 1, 1)
        # dgrad operands: bf16(W * scale[co]) -- the BatchNorm scale is folded BEFORE the storage rounding
        wT3, wT2, wT1 = self._Rw(w3 * v(s3)), self._Rw(w2 * v(s2)), self._Rw(w1 * v(s1))
        dU2 = self._R(conv2d_input(a2.shape, wT3, dC) * (a2 > 0), bi)
        dU1 = self._R(conv2d_input(a1.shape, wT2, dU2, 1, dil, dil) * (a1 > 0), bi)
        acc(pre + '.conv3.weight', conv2d_weight(a2, w3.shape, dC) * v(s3))
        acc(pre + '.conv2.weight', conv2d_weight(a1, w2.shape, dU2, 1, dil, dil) * v(s2))
        if down:
            sd, _ = self._affine(pre + '.downsample.1')
            wd = self._Rw(st[pre + '.downsample.0.weight'])
            acc(pre + '.downsample.0.weight', conv2d_weight(xin, wd.shape, dC, stride) * v(sd))
            dres = self._R(conv2d_input(xin.shape, self._Rw(wd * v(sd)), dC, stride), bi)
        else:
            dres = dC
        acc(pre + '.conv1.weight', conv2d_weight(xin, w1.shape, dU1, stride) * v(s1))
        t = conv2d_input(xin.shape, wT1, dU1, stride) + dres
        if bi > 0:
            t = t * (xin > 0)             # (block 0: the stem's ReLU mask is applied by the max-pool backward)
        return self._R(t, bi)

    def stem_backward(self, x, s, dp, acc):
        """max-pool backward (ties: first maximum wins, like ATen) fused with the ReLU mask, then dW of conv1."""
        with torch.enable_grad():
            sl = s.detach().requires_grad_(True)
            F.max_pool2d(sl, 3, 2, 1, ceil_mode=True).backward(dp)
        ds = self._R(sl.grad * (s > 0))
        sc, _ = self._affine('bn1')
        acc('conv1.weight', conv2d_weight(x, self.st['conv1.weight'].shape, ds, 2, 3) * sc.view(-1, 1, 1, 1))

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, saved, dlogits, grads=None):
        """dlogits fp32 (N,C,h,w) -> dict key -> gradient (accumulated into `grads` when given). Weight gradients stay
        fp32 (the device accumulates them in an fp32 arena)."""
        g = grads if grads is not None else OrderedDict()

        def acc(key, val):
            g[key] = val if key not in g else g[key] + val

        with torch.no_grad():
            dC = self.head_backward(saved['x4'], dlogits, acc)
            for bi in range(len(self.units) - 1, -1, -1):
                xin, a1, a2 = saved['blocks'][bi]
                dC = self.block_backward(bi, xin, a1, a2, dC, acc)
            self.stem_backward(saved['x'], saved['s'], dC, acc)
        return g


def forward_lowres(x, st, num_classes, layers=dl.LAYERS, storage='bf16'):
    return Chain(st, num_classes, layers, storage).forward(x, save=False)[0]
