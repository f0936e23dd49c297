"""
Flat fp32 parameter arenas.

The reference keeps 528 separate float state tensors per network and walks them from Python for the optimizer
(314 + 8 entries) and the EMA (528 x 3 launches), optim_weight_ema.py:21-25 / train_seg_semisup_mask_mt.py:465-467.
Here every float tensor of a module's state_dict (parameters AND buffers, in state_dict order, each aligned to 64
elements) is re-homed into ONE contiguous fp32 CUDA buffer; the module's tensors become views into it. That gives

  * one fused optimizer + EMA launch per step (csrc/optim.hip),
  * one gradient buffer (`grad`) -> a single large all-reduce over RCCL instead of per-tensor buckets,
  * bf16 copies of student / teacher weights written by the same kernel for the bf16 conv path.

Convolution weights (4-D tensors, logical shape (Cout, Cin, kh, kw) as in the reference's state_dict) are stored
PHYSICALLY as [kh][kw][Cout][Cin]: the module sees a strided view with the reference's shape, while the flat
buffers hold exactly the [tap][Cout][Cin] operand layout of the MFMA convolution kernels (csrc/conv.hip). The bf16
copy written by the optimizer kernel is therefore directly the forward weight operand, and the weight-gradient
kernel accumulates straight into the fp32 gradient arena. Element-wise consumers (Adam, EMA, all-reduce) do not care.

int64 `num_batches_tracked` buffers are left alone (the reference's EMA never touches them).
"""
from collections import OrderedDict

import numpy as np
import torch

ALIGN = 64   # elements (256 B)


class Segment(object):
    __slots__ = ('key', 'offset', 'count', 'shape', 'is_param', 'requires_grad')

    def __init__(self, key, offset, count, shape, is_param, requires_grad):
        self.key, self.offset, self.count, self.shape = key, offset, count, tuple(shape)
        self.is_param, self.requires_grad = is_param, requires_grad


def _logical_view(flat_slice, shape):
    """View of a flat segment with logical `shape`; 4-D tensors use the physical order (kh, kw, co, ci)."""
    if len(shape) == 4:
        co, ci, kh, kw = shape
        return flat_slice.view(kh, kw, co, ci).permute(2, 3, 0, 1)
    return flat_slice.view(shape)


class ParamArena(object):
    """Re-homes the float32 state of `module` into one flat CUDA buffer (in place; tensor values preserved)."""

    def __init__(self, module, with_grad=True, with_bf16=False):
        self.module = module
        named = OrderedDict()
        params = dict(module.named_parameters())
        for key, t in module.state_dict(keep_vars=True).items():
            if t.dtype == torch.float32:
                named[key] = t
        if not named:
            raise ValueError('module has no float32 state')
        dev = next(iter(named.values())).device
        if dev.type != 'cuda':
            raise RuntimeError('ParamArena needs the module on the GPU (got {}); no CPU fallback'.format(dev))
        self.device = dev
        self.segments = []
        off = 0
        for key, t in named.items():
            cnt = t.numel()
            self.segments.append(Segment(key, off, cnt, t.shape, key in params,
                                         bool(key in params and params[key].requires_grad)))
            off += (cnt + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        # bumped by everything of this package that writes the weights (fused optimizer / EMA steps, refresh_bf16 behind
        # load_state_dict): derived operands cached per layer (zero-padded weights, transposed data-gradient operands of the general
        # convolution path, backbone_hip._HipConvGeneralFn) are valid for one version
        self.version = 0
        self.derived = {}            # (key, what, dtype) -> (version, tensor)
        self._params = None          # [(segment, parameter)] of the trainable segments, built on first use
        self.by_key = {s.key: s for s in self.segments}
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev) if with_grad else None
        self.bf16 = torch.zeros(self.total, dtype=torch.bfloat16, device=dev) if with_bf16 else None
        with torch.no_grad():
            for s in self.segments:
                t = named[s.key]
                view = _logical_view(self.flat[s.offset:s.offset + s.count], s.shape)
                view.copy_(t)
                t.data = view            # parameters and buffers alike keep their identity, storage moves
                if with_grad and s.requires_grad:
                    t.grad = _logical_view(self.grad[s.offset:s.offset + s.count], s.shape)
        if with_bf16:
            self.refresh_bf16()

    def view(self, key, buf=None):
        """Logical-shape view of `key` in `buf` (default: the fp32 master arena)."""
        s = self.by_key[key]
        buf = self.flat if buf is None else buf
        return _logical_view(buf[s.offset:s.offset + s.count], s.shape)

    def packed(self, key, buf=None):
        """Physical (ntaps, Cout, Cin) view of a convolution weight in `buf` -- the kernels' operand layout."""
        s = self.by_key[key]
        co, ci, kh, kw = s.shape
        buf = self.flat if buf is None else buf
        return buf[s.offset:s.offset + s.count].view(kh * kw, co, ci)

    def refresh_bf16(self):
        self.version += 1
        if self.bf16 is not None:
            self.bf16.copy_(self.flat)

    def touch(self):
        """The weights were written (optimizer / EMA step): cached derived operands are stale."""
        self.version += 1

    def cached(self, key, what, dtype, make):
        """`make()` once per weight version: a derived operand of layer `key` (padded / transposed copies)."""
        k = (key, what, dtype)
        hit = self.derived.get(k)
        if hit is not None and hit[0] == self.version:
            return hit[1]
        t = make()
        self.derived[k] = (self.version, t)
        return t

    def zero_grad(self):
        if self.grad is not None:
            self.grad.zero_()

    def ensure_grads_attached(self):
        """`module.zero_grad()` (set_to_none) drops the parameters' `.grad` views of the gradient arena. The kernels
        accumulate into the arena regardless; before they do, give the views back -- and since "None" means "zero" to
        the caller, clear what the arena still holds for those parameters."""
        if self.grad is None:
            return
        if self._params is None:
            params = dict(self.module.named_parameters())
            self._params = [(s, params[s.key]) for s in self.segments if s.requires_grad]
        base = self.grad.data_ptr()
        for s, p in self._params:
            g = p.grad
            if g is None:
                self.grad[s.offset:s.offset + s.count].zero_()
                p.grad = _logical_view(self.grad[s.offset:s.offset + s.count], s.shape)
            elif g.data_ptr() != base + 4 * s.offset:
                # a foreign gradient tensor (autograd made it while the view was gone): fold it in, then re-home
                view = _logical_view(self.grad[s.offset:s.offset + s.count], s.shape)
                view.add_(g.to(view.dtype))
                p.grad = view

    def keys(self):
        return [s.key for s in self.segments]

    def same_layout(self, other):
        return (self.total == other.total and len(self.segments) == len(other.segments) and
                all(a.key == b.key and a.offset == b.offset and a.count == b.count
                    for a, b in zip(self.segments, other.segments)))


def arena_of(module):
    return getattr(module, '_cms_arena', None)


def ensure_arena(module, with_grad=True, with_bf16=False):
    a = arena_of(module)
    if a is None:
        a = ParamArena(module, with_grad=with_grad, with_bf16=with_bf16)
        module._cms_arena = a
    else:
        if with_bf16 and a.bf16 is None:
            a.bf16 = torch.zeros(a.total, dtype=torch.bfloat16, device=a.device)
            a.refresh_bf16()
        if with_grad and a.grad is None:
            a.grad = torch.zeros(a.total, dtype=torch.float32, device=a.device)
            for s in a.segments:
                if s.requires_grad:
                    dict(module.named_parameters())[s.key].grad = _logical_view(
                        a.grad[s.offset:s.offset + s.count], s.shape)
    return a


def build_chunk_table(segments_kcount, chunk):
    """[(count), ...] per segment -> (chunk_seg uint32[], chunk_off uint32[]) covering every element once."""
    seg_ids, offs = [], []
    for i, cnt in enumerate(segments_kcount):
        n = (cnt + chunk - 1) // chunk
        seg_ids.append(np.full(n, i, dtype=np.uint32))
        offs.append(np.arange(n, dtype=np.uint32) * np.uint32(chunk))
    return np.concatenate(seg_ids), np.concatenate(offs)
