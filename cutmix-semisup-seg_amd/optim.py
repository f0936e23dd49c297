"""
Fused Adam / SGD (+ teacher EMA) over the flat parameter arena -- the device side of
`torch.optim.Adam([...pretrained @ 0.1 lr..., ...new @ lr...])` / `torch.optim.SGD(...)` in
train_seg_semisup_mask_mt.py:90-100 and of `student_optim.step(); teacher_optim.step()` at :465-467.

Semantics kept from the reference:
  * parameter groups with their own `lr` (and `initial_lr`, so lr_schedules work on `param_groups`);
  * a tensor listed k times in a group (architectures/deeplab2.py:208-230 yields backbone conv weights 3x / 4x)
    receives k sequential updates per step and Adam's step counter advances by k;
  * parameters that never receive a gradient (DeepLab v2's ASPP d18 / d24 branches, Q1) are not updated;
  * a tensor may not sit in two groups (torch raises ValueError).
One kernel launch per step for all parameters; learning rates travel as a small device array so the launch is
hipGraph-replayable; the step counter lives on the device.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from ._lib import fn, check
from .arena import ensure_arena, build_chunk_table


def ops_join_side_streams():
    """Weight gradients the layer engines issued on side streams (ops.layer_wgrad_stream) must have landed before the gradient arena
    is read or cleared on the current stream."""
    from . import ops
    ops.join_side_streams()


class _FusedOptimizer(object):
    KIND = None
    LR_RING = 16

    def __init__(self, module, param_groups, defaults):
        self.module = module
        self.arena = ensure_arena(module, with_grad=True)
        self.defaults = dict(defaults)
        a = self.arena
        ptr2seg = {}
        base = a.flat.data_ptr()
        for i, s in enumerate(a.segments):
            ptr2seg[base + 4 * s.offset] = i
        self.param_groups = []
        mult = [0] * len(a.segments)
        group_of = [-1] * len(a.segments)
        for gi, g in enumerate(param_groups):
            g = dict(g)
            plist = list(g['params'])
            g['params'] = plist
            for k, v in self.defaults.items():
                g.setdefault(k, v)
            g.setdefault('initial_lr', g['lr'])
            self.param_groups.append(g)
            for p in plist:
                si = ptr2seg.get(p.data_ptr())
                if si is None:
                    raise ValueError('optimizer got a parameter that does not live in the module\'s arena')
                if group_of[si] not in (-1, gi):
                    raise ValueError('some parameters appear in more than one parameter group')
                group_of[si] = gi
                mult[si] += 1
        unused = set(getattr(module, 'unused_parameter_keys', lambda: [])())
        if max(mult) > 8:
            raise ValueError('a parameter is listed more than 8 times')
        segs = (_lib.ParamSegment * len(a.segments))()
        for i, s in enumerate(a.segments):
            segs[i].offset = s.offset
            segs[i].count = s.count
            k = mult[i] if (s.requires_grad and s.key not in unused) else 0
            segs[i].k_updates = k
            segs[i].lr_group = max(group_of[i], 0)
        self.k_updates = OrderedDict((s.key, int(segs[i].k_updates)) for i, s in enumerate(a.segments))
        dev = a.device
        self._segments = torch.frombuffer(bytearray(bytes(segs)), dtype=torch.uint8).to(dev)
        cs, co = build_chunk_table([s.count for s in a.segments], _lib.OPT_CHUNK)
        self._chunk_seg = torch.from_numpy(cs.view(np.int32)).to(dev)
        self._chunk_off = torch.from_numpy(co.view(np.int32)).to(dev)
        self._n_chunks = int(cs.shape[0])
        # first chunk of the segment that starts at a given arena element (+ the end of the arena): ranged launches
        self._chunk_at = {}
        c0 = 0
        for s in a.segments:
            self._chunk_at[int(s.offset)] = c0
            c0 += (int(s.count) + _lib.OPT_CHUNK - 1) // _lib.OPT_CHUNK
        self._chunk_at[int(a.flat.numel())] = c0
        self._early = None                  # chunk ranges already updated in this step (begin_ranged / step_range)
        self.slot0 = torch.zeros_like(a.flat)
        self.slot1 = torch.zeros_like(a.flat) if self.KIND == 'adam' else None
        # learning rates travel host -> device through a RING of pinned slots, each guarded by an event: the trainer
        # never syncs per iteration, so the host may run whole iterations ahead of the GPU and must not rewrite a
        # slot whose asynchronous copy has not executed yet (a single slot would let step i run with the lr of i+1)
        ng = max(len(self.param_groups), 1)
        self._lr_ring = [torch.zeros(ng, dtype=torch.float64).pin_memory() for _ in range(self.LR_RING)]
        self._lr_events = [None] * self.LR_RING
        self._lr_slot = 0
        self._lrs_last = None
        self._lrs_dev = torch.zeros(ng, dtype=torch.float64, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.grad_scale = 1.0
        self._ema = None

    # -- torch.optim-like surface
    def zero_grad(self, set_to_none=False):
        """Clears the gradient arena. The fused kernels read gradients from the ARENA only, so every parameter's `.grad`
        must be its view of it: if somebody dropped the views (`module.zero_grad(set_to_none=True)`, `p.grad = None`),
        autograd would create fresh `.grad` tensors outside the arena for the parameters it owns (head, BatchNorm,
        decoder layers) and their updates would be silently lost -- the views are re-homed here, at the start of
        every step, over ALL segments."""
        ops_join_side_streams()
        self.arena.ensure_grads_attached()
        self.arena.zero_grad()

    def attach_ema(self, ema_optimizer):
        if not ema_optimizer.target_arena.same_layout(self.arena):
            raise ValueError('teacher and student arenas differ')
        self._ema = ema_optimizer

    def state_dict(self):
        return dict(kind=self.KIND, step_count=int(self.step_count.item()), slot0=self.slot0, slot1=self.slot1,
                    param_groups=[{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups])

    def _desc(self):
        a = self.arena
        d = _lib.OptimDesc()
        d.param, d.grad = a.flat.data_ptr(), a.grad.data_ptr()
        d.slot0 = self.slot0.data_ptr()
        d.slot1 = self.slot1.data_ptr() if self.slot1 is not None else None
        d.param_bf16 = a.bf16.data_ptr() if a.bf16 is not None else None
        if self._ema is not None:
            t = self._ema.target_arena
            d.ema_param = t.flat.data_ptr()
            d.ema_bf16 = t.bf16.data_ptr() if t.bf16 is not None else None
            alpha = float(self._ema.ema_alpha)
            d.ema_alpha = alpha
            d.ema_one_minus_alpha = 1.0 - alpha
        else:
            d.ema_param = None
            d.ema_bf16 = None
        d.segments = self._segments.data_ptr()
        d.chunk_seg = self._chunk_seg.data_ptr()
        d.chunk_off = self._chunk_off.data_ptr()
        d.n_chunks = self._n_chunks
        d.lrs = self._lrs_dev.data_ptr()
        d.step_count = self.step_count.data_ptr()
        d.grad_scale = float(self.grad_scale)
        return d

    def _fill(self, d):
        raise NotImplementedError

    def _upload_lrs(self):
        lrs = tuple(float(g['lr']) for g in self.param_groups)
        if lrs == self._lrs_last:
            return                                   # unchanged since the last upload ('none' schedule, stepped epochs)
        k = self._lr_slot
        if self._lr_events[k] is not None:
            self._lr_events[k].synchronize()         # LR_RING uploads ago: practically never blocks
        host = self._lr_ring[k]
        for i, v in enumerate(lrs):
            host[i] = v
        self._lrs_dev.copy_(host, non_blocking=True)
        ev = self._lr_events[k] or torch.cuda.Event()
        ev.record()
        self._lr_events[k] = ev
        self._lr_slot = (k + 1) % self.LR_RING
        self._lrs_last = lrs

    def _check_uniform(self, names):
        """The kernels take these hyper-parameters once per launch: per-group overrides are an error, not silently
        group 0's value."""
        for name in names:
            vals = [g[name] for g in self.param_groups]
            if any(v != vals[0] for v in vals[1:]):
                raise ValueError('{}: parameter groups differ in `{}` ({}); only `lr` may differ per group'.format(
                    type(self).__name__, name, vals))

    def _launch(self, first_chunk, n_chunks):
        if n_chunks <= 0:
            return
        ops_join_side_streams()           # weight gradients the layer engines put on side streams (ops.layer_wgrad_stream)
        d = self._desc()
        self._fill(d)
        d.chunk_seg = self._chunk_seg.data_ptr() + 4 * first_chunk
        d.chunk_off = self._chunk_off.data_ptr() + 4 * first_chunk
        d.n_chunks = int(n_chunks)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(fn['cms_adam_ema_step' if self.KIND == 'adam' else 'cms_sgd_ema_step'](C.byref(d), stream),
              'cms_{}_ema_step'.format(self.KIND))

    # -- ranged launches: parts of the step issued EARLY, as soon as their gradients are final (step.py: the backward pass
    # finishes the arena from the end towards the start, the update of [layer4 + head] can run while layer1's gradients are
    # still being computed). `begin_ranged()` on the stream the backward pass starts from (learning rates uploaded there),
    # `step_range(lo, hi)` for arena elements [lo, hi) (segment-aligned) on the stream that holds their final gradients,
    # `step()` then updates what is left and closes the step (one step-counter increment for all of it).
    def begin_ranged(self):
        self._upload_lrs()
        self._early = []

    def step_range(self, lo, hi):
        if self._early is None:
            raise RuntimeError('step_range() outside begin_ranged() .. step()')
        c0, c1 = self._chunk_at[int(lo)], self._chunk_at[int(hi)]
        for a0, a1 in self._early:
            if c0 < a1 and a0 < c1:
                raise RuntimeError('step_range: elements updated twice in one step')
        self._launch(c0, c1 - c0)
        self._early.append((c0, c1))

    def step(self):
        self._upload_lrs()
        done = sorted(self._early or [])
        self._early = None
        pos = 0
        for c0, c1 in done + [(self._n_chunks, self._n_chunks)]:       # the complement of the early ranges
            self._launch(pos, c0 - pos)
            pos = max(pos, c1)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(fn['cms_increment_counter'](C.c_void_p(self.step_count.data_ptr()), stream), 'cms_increment_counter')
        from .backbone_hip import executors_of
        self.arena.touch()
        for ex in executors_of(self.module):
            ex.weights_changed()           # packed backward weights are stale now
        if self._ema is not None:
            self._ema._touch_target()
        if self._ema is not None:
            self._ema._mark_fused_step_done()


class FusedAdam(_FusedOptimizer):
    KIND = 'adam'

    def __init__(self, module, param_groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super(FusedAdam, self).__init__(module, param_groups, dict(lr=lr, betas=betas, eps=eps))

    def _fill(self, d):
        self._check_uniform(('betas', 'eps'))
        g = self.param_groups[0] if self.param_groups else self.defaults
        d.beta1, d.beta2 = float(g['betas'][0]), float(g['betas'][1])
        d.eps = float(g['eps'])


class FusedSGD(_FusedOptimizer):
    KIND = 'sgd'

    def __init__(self, module, param_groups, lr=1e-3, momentum=0.0, nesterov=False, weight_decay=0.0):
        super(FusedSGD, self).__init__(module, param_groups, dict(lr=lr, momentum=momentum, nesterov=nesterov,
                                                                  weight_decay=weight_decay))

    def _fill(self, d):
        # Duplicated entries (deeplab2.py:208-230) follow torch >= 1.5 semantics -- the torch of this image, 2.10, which
        # the oracle (oracle/ema_opt.py:sgd_k_updates) pins: on the very first step every visit of a tensor starts its
        # momentum buffer from the gradient. The reference environment's torch 1.4 looked the buffer up per visit, so
        # its 2nd / 3rd visit of step 0 already saw one ("Unpinned").
        self._check_uniform(('momentum', 'weight_decay', 'nesterov'))
        g = self.param_groups[0] if self.param_groups else self.defaults
        d.momentum = float(g['momentum'])
        d.weight_decay = float(g['weight_decay'])
        d.nesterov = int(bool(g['nesterov']))
