"""
Tensor-level wrappers over the C ABI (include/cutmixseg.h). PyTorch is used for device memory, streams and
autograd plumbing only; all arithmetic happens in the HIP kernels. Every function requires CUDA(HIP) tensors and
raises otherwise -- there is no CPU path.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import fn, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('cutmix-semisup-seg_amd ops run on the GPU only (got a {} tensor); there is no CPU '
                               'fallback'.format(t.device))


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _dtype_code(t):
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.bfloat16:
        return _lib.BF16
    raise TypeError('unsupported dtype {}'.format(t.dtype))


# ---------------------------------------------------------------------------------------------- box masks / paste
def ranges_to_device(ranges, device):
    """int32 (N, n_boxes, 4) numpy / tensor -> contiguous int32 CUDA tensor."""
    if isinstance(ranges, np.ndarray):
        ranges = torch.from_numpy(np.ascontiguousarray(ranges, dtype=np.int32))
    if not ranges.is_cuda and torch.device(device).type == 'cuda':
        # through PINNED memory: a "non_blocking" copy from pageable memory is a blocking one (the runtime stages it when the
        # stream gets there), i.e. the host would wait for the whole previous iteration at the first line of every iteration
        # and the GPU would then idle through the host's launch latencies (1 ms of a 19.9 ms step: profiles/r04k_*). The
        # caching host allocator keeps the staging block alive until the copy has run.
        ranges = ranges.to(torch.int32).contiguous().pin_memory()
    return ranges.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()


def boxmask_rasterize(ranges, mask_shape, invert, out=None):
    """ranges int32 CUDA (N, nb, 4) -> f32 (N,1,H,W). mask_gen.py:110-116 on the device."""
    _need_cuda(ranges)
    n, nb = int(ranges.shape[0]), int(ranges.shape[1])
    H, W = int(mask_shape[0]), int(mask_shape[1])
    if out is None:
        out = torch.empty((n, 1, H, W), dtype=torch.float32, device=ranges.device)
    check(fn['cms_boxmask_rasterize'](_ptr(ranges), n, nb, H, W, int(bool(invert)), _ptr(out), _stream()),
          'cms_boxmask_rasterize')
    return out


def cutmix_paste(x0, x1, ranges=None, invert=True, mask=None, out=None):
    """
    out = x0*(1-m) + x1*m  (train_seg_semisup_mask_mt.py:350-351, 363); x0=None gives x1*m (:389).
    Give either int32 `ranges` (mask rasterised in-kernel) or a materialised f32 `mask` (N,1,H,W).
    """
    _need_cuda(x0, x1, ranges, mask)
    x1 = x1.contiguous()
    if x0 is not None:
        x0 = x0.contiguous()
        if x0.shape != x1.shape or x0.dtype != x1.dtype:
            raise ValueError('cutmix_paste: x0/x1 shape or dtype mismatch')
    n, c, h, w = (int(s) for s in x1.shape)
    if out is None:
        out = torch.empty_like(x1)
    dt = _dtype_code(x1)
    if (ranges is None) == (mask is None):
        raise ValueError('cutmix_paste: give exactly one of ranges / mask')
    if ranges is not None:
        check(fn['cms_cutmix_paste'](_ptr(x0), _ptr(x1), _ptr(out), dt, _ptr(ranges), n, int(ranges.shape[1]), c, h, w,
                                     int(bool(invert)), _stream()), 'cms_cutmix_paste')
    else:
        mask = _f32c(mask)
        check(fn['cms_cutmix_paste_mask'](_ptr(x0), _ptr(x1), _ptr(out), dt, _ptr(mask), n, c, h, w, _stream()),
              'cms_cutmix_paste_mask')
    return out


# ---------------------------------------------------------------------------------------------- consistency loss
class ConsistencyConfig(object):
    """Static configuration of the unsupervised loss (train_seg_semisup_mask_mt.py CLI flags)."""

    def __init__(self, mode='mix', loss_fn='var', conf_thresh=0.97, conf_per_pixel=False, align_corners=True,
                 invert=True):
        if mode not in ('mix', 'cut'):
            raise ValueError('Unknown mask_mode {}'.format(mode))
        if loss_fn not in _lib.LOSS_IDS:
            raise ValueError('Unknown consistency loss function {}'.format(loss_fn))
        self.mode = mode
        self.loss_fn = loss_fn
        self.conf_thresh = float(conf_thresh)
        self.conf_per_pixel = bool(conf_per_pixel)
        self.align_corners = bool(align_corners)
        self.invert = bool(invert)


def _cons_desc(cfg, l_stu, l_tea0, l_tea1, ranges, mask, um0, um1, out_size):
    n, c, h, w = (int(s) for s in l_stu.shape)
    d = _lib.ConsistencyDesc()
    d.l_stu, d.l_tea0, d.l_tea1 = l_stu.data_ptr(), l_tea0.data_ptr(), (l_tea1.data_ptr() if l_tea1 is not None else None)
    d.ranges = ranges.data_ptr() if ranges is not None else None
    d.mask = mask.data_ptr() if mask is not None else None
    d.um0 = um0.data_ptr() if um0 is not None else None
    d.um1 = um1.data_ptr() if um1 is not None else None
    d.n, d.c, d.h, d.w = n, c, h, w
    d.H, d.W = int(out_size[0]), int(out_size[1])
    d.align_corners = int(cfg.align_corners)
    d.n_boxes = int(ranges.shape[1]) if ranges is not None else 0
    d.invert = int(cfg.invert)
    d.mode = _lib.MODE_MIX if cfg.mode == 'mix' else _lib.MODE_CUT
    d.loss_fn = _lib.LOSS_IDS[cfg.loss_fn]
    d.conf_thresh = cfg.conf_thresh
    d.conf_per_pixel = int(cfg.conf_per_pixel)
    return d


def _allreduce_sum(t, group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return True
    return False


def consistency_forward(cfg, l_stu, l_tea0, l_tea1, out_size, ranges=None, mask=None, um0=None, um1=None,
                        ramp_val=1.0, cons_weight=1.0, group=None, sync_conf_rate=True):
    """
    Fused forward. Returns (scalars, ctx): scalars = f32[4] device tensor
    [consistency_loss, conf_rate, grad_scale, unsup_loss]; ctx feeds consistency_backward. No host sync.
    Under torch.distributed the confidence count is all-reduced so that the rate is the global one.
    """
    _need_cuda(l_stu, l_tea0, l_tea1, ranges, mask, um0, um1)
    l_stu, l_tea0, l_tea1 = _f32c(l_stu), _f32c(l_tea0), _f32c(l_tea1)
    mask, um0, um1 = _f32c(mask), _f32c(um0), _f32c(um1)
    if l_tea0.shape != l_stu.shape or (l_tea1 is not None and l_tea1.shape != l_stu.shape):
        raise ValueError('consistency: student / teacher logits shapes differ')
    d = _cons_desc(cfg, l_stu, l_tea0, l_tea1, ranges, mask, um0, um1, out_size)
    dev = l_stu.device
    ws = torch.empty(max(int(fn['cms_consistency_workspace_bytes'](C.byref(d))), 16), dtype=torch.uint8, device=dev)
    stats = torch.empty(4, dtype=torch.float64, device=dev)
    check(fn['cms_consistency_fwd'](C.byref(d), _ptr(ws), _ptr(stats), _stream()), 'cms_consistency_fwd')
    stats_g = stats
    if sync_conf_rate and cfg.conf_thresh > 0.0:
        g = stats.clone()
        if _allreduce_sum(g, group):
            stats_g = g
    scalars = torch.empty(4, dtype=torch.float32, device=dev)
    check(fn['cms_consistency_finalize'](_ptr(stats), _ptr(stats_g), cfg.conf_thresh, int(cfg.conf_per_pixel),
                                         float(ramp_val), float(cons_weight), _ptr(scalars), _stream()),
          'cms_consistency_finalize')
    keep = (l_stu, l_tea0, l_tea1, ranges, mask, um0, um1, cfg, tuple(int(v) for v in out_size))
    return scalars, (d, keep, stats)


def consistency_backward(ctx, scalars, grad_out=None, samples=None):
    """grad wrt the (low-res) student logits; `grad_out` f32 (N,C,h,w) is accumulated into when given. `samples` = (s0, s1): only
    that run of samples (rows s0:s1 of `grad_out` are written) -- the per-pixel work of the backward is independent between samples,
    so a caller may issue disjoint runs on different streams (step.py: half of it on the main stream beside the other half)."""
    d, keep, _ = ctx
    l_stu = keep[0]
    if grad_out is None:
        grad_out = torch.zeros_like(l_stu)
    if samples is not None:
        s0, s1 = int(samples[0]), int(samples[1])
        if not (0 <= s0 < s1 <= int(l_stu.shape[0])):
            raise ValueError('consistency_backward: bad sample range')
        sl = lambda t: None if t is None else t[s0:s1]
        l_s, l_t0, l_t1, ranges, mask, um0, um1, cfg, out_size = keep
        d = _cons_desc(cfg, sl(l_s), sl(l_t0), sl(l_t1), sl(ranges), sl(mask), sl(um0), sl(um1), out_size)
        check(fn['cms_consistency_bwd'](C.byref(d), _ptr(scalars), _ptr(grad_out[s0:s1]), _stream()), 'cms_consistency_bwd')
        return grad_out
    check(fn['cms_consistency_bwd'](C.byref(d), _ptr(scalars), _ptr(grad_out), _stream()), 'cms_consistency_bwd')
    return grad_out


def consistency_fused(cfg, l_stu, l_tea0, l_tea1, out_size, grad_out, ranges=None, mask=None, um0=None, um1=None,
                      ramp_val=1.0, cons_weight=1.0, group=None, sync_conf_rate=True):
    """(round 6) `consistency_forward` + `consistency_backward` as ONE loss launch (cms_consistency_fwd_bwd): returns the same
    scalars and ADDS d unsup_loss / d l_stu to `grad_out` (f32 (N,C,h,w), its rows must be ZERO on entry: the deferred scalar factor
    -- the confidence rate of the default mode, in which the gradient is linear -- is applied to the rows afterwards). Falls back to
    the two launches where the fused one does not exist (identity geometry, deterministic mode, CMS_LOSS_FUSED=0)."""
    _need_cuda(l_stu, l_tea0, l_tea1, ranges, mask, um0, um1, grad_out)
    l_stu, l_tea0, l_tea1 = _f32c(l_stu), _f32c(l_tea0), _f32c(l_tea1)
    mask, um0, um1 = _f32c(mask), _f32c(um0), _f32c(um1)
    if l_tea0.shape != l_stu.shape or (l_tea1 is not None and l_tea1.shape != l_stu.shape):
        raise ValueError('consistency: student / teacher logits shapes differ')
    if grad_out.dtype != torch.float32 or tuple(grad_out.shape) != tuple(l_stu.shape) or not grad_out.is_contiguous():
        raise ValueError('consistency_fused: grad_out must be a contiguous f32 tensor of the logits\' shape')
    d = _cons_desc(cfg, l_stu, l_tea0, l_tea1, ranges, mask, um0, um1, out_size)
    if not fn['cms_consistency_fused_supported'](C.byref(d)):
        sc, cctx = consistency_forward(cfg, l_stu, l_tea0, l_tea1, out_size, ranges, mask, um0, um1, ramp_val, cons_weight, group,
                                       sync_conf_rate)
        consistency_backward(cctx, sc, grad_out)
        return sc
    dev = l_stu.device
    ws = torch.empty(max(int(fn['cms_consistency_workspace_bytes'](C.byref(d))), 16), dtype=torch.uint8, device=dev)
    stats = torch.empty(4, dtype=torch.float64, device=dev)
    P = float(d.n) * float(d.H) * float(d.W)
    grad_unit = float(ramp_val) * float(cons_weight) / P
    check(fn['cms_consistency_fwd_bwd'](C.byref(d), grad_unit, _ptr(ws), _ptr(stats), _ptr(grad_out), _stream()),
          'cms_consistency_fwd_bwd')
    stats_g = stats
    if sync_conf_rate and cfg.conf_thresh > 0.0:
        g = stats.clone()
        if _allreduce_sum(g, group):
            stats_g = g
    scalars = torch.empty(4, dtype=torch.float32, device=dev)
    check(fn['cms_consistency_finalize'](_ptr(stats), _ptr(stats_g), cfg.conf_thresh, int(cfg.conf_per_pixel),
                                         float(ramp_val), float(cons_weight), _ptr(scalars), _stream()),
          'cms_consistency_finalize')
    if cfg.conf_thresh > 0.0 and not cfg.conf_per_pixel:
        # default confidence mode (:415-418): the loss mask is the scalar RATE -- the factor the launch above left out
        check(fn['cms_scale_by_scalar'](_ptr(grad_out), grad_out.numel(), _ptr(scalars), 1, 1.0, _stream()), 'cms_scale_by_scalar')
    return scalars


class _ConsistencyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, l_stu, l_tea0, l_tea1, ranges, mask, um0, um1, cfg, out_size, ramp_val, cons_weight, group):
        scalars, c = consistency_forward(cfg, l_stu.detach(), l_tea0, l_tea1, out_size, ranges, mask, um0, um1,
                                         ramp_val, cons_weight, group)
        ctx.c = c
        ctx.scalars = scalars
        ctx.in_dtype = l_stu.dtype
        closs, rate, unsup = scalars[0], scalars[1], scalars[3]
        ctx.mark_non_differentiable(closs, rate)
        return unsup, closs, rate

    @staticmethod
    def backward(ctx, g_unsup, g_closs, g_rate):
        sc = ctx.scalars.clone()
        sc[2] = sc[2] * g_unsup
        grad = consistency_backward(ctx.c, sc)
        return (grad.to(ctx.in_dtype),) + (None,) * 11


def consistency_loss(l_stu, l_tea0, l_tea1, out_size, cfg, ranges=None, mask=None, um0=None, um1=None, ramp_val=1.0,
                     cons_weight=1.0, group=None):
    """autograd entry: returns (unsup_loss [differentiable wrt l_stu], consistency_loss, conf_rate)."""
    return _ConsistencyFn.apply(l_stu, l_tea0, l_tea1, ranges, mask, um0, um1, cfg, tuple(out_size), ramp_val,
                                cons_weight, group)


# ---------------------------------------------------------------------------------------------- supervised CE
def _ce_desc(logits, labels, ignore_index, out_size, align_corners):
    n, c, h, w = (int(s) for s in logits.shape)
    d = _lib.CeDesc()
    d.logits, d.labels = logits.data_ptr(), labels.data_ptr()
    if labels.dtype == torch.uint8:
        d.label_dtype = _lib.LABEL_U8
    elif labels.dtype == torch.int64:
        d.label_dtype = _lib.LABEL_I64
    else:
        raise TypeError('labels must be uint8 or int64, got {}'.format(labels.dtype))
    d.ignore_index = int(ignore_index)
    d.n, d.c, d.h, d.w = n, c, h, w
    d.H, d.W = int(out_size[0]), int(out_size[1])
    d.align_corners = int(bool(align_corners))
    return d


def ce_forward(logits, labels, out_size=None, ignore_index=255, align_corners=True, loss_weight=1.0, group=None,
               sync_count=False):
    """labels (N,H,W) uint8/int64. Returns (scalars f32[2] = [loss, grad_scale], ctx)."""
    _need_cuda(logits, labels)
    logits = _f32c(logits)
    labels = labels.contiguous()
    if labels.dim() == 4:
        labels = labels[:, 0].contiguous()
    if out_size is None:
        out_size = labels.shape[1:3]
    if tuple(labels.shape) != (logits.shape[0], int(out_size[0]), int(out_size[1])):
        raise ValueError('ce: labels shape {} does not match (N,H,W)=({}, {}, {})'.format(
            tuple(labels.shape), logits.shape[0], out_size[0], out_size[1]))
    d = _ce_desc(logits, labels, ignore_index, out_size, align_corners)
    dev = logits.device
    ws = torch.empty(max(int(fn['cms_ce_workspace_bytes'](C.byref(d))), 16), dtype=torch.uint8, device=dev)
    stats = torch.empty(2, dtype=torch.float64, device=dev)
    check(fn['cms_ce_fwd'](C.byref(d), _ptr(ws), _ptr(stats), _stream()), 'cms_ce_fwd')
    if sync_count:
        # exact global-batch semantics under data parallelism: divide by the mean valid count over ranks
        import torch.distributed as dist
        cnt = stats[1:2].clone()
        if _allreduce_sum(cnt, group):
            stats = torch.stack([stats[0], cnt[0] / dist.get_world_size(group)])
    scalars = torch.empty(2, dtype=torch.float32, device=dev)
    check(fn['cms_ce_finalize'](_ptr(stats), float(loss_weight), _ptr(scalars), _stream()), 'cms_ce_finalize')
    return scalars, (d, (logits, labels), stats)


def ce_fused(logits, labels, grad_out, out_size=None, ignore_index=255, align_corners=True, loss_weight=1.0, group=None,
             sync_count=False):
    """(round 6) `ce_forward` + `ce_backward` as ONE loss launch (cms_ce_fwd_bwd): returns the scalars f32[2] = [loss, grad_scale]
    and ADDS the gradient to `grad_out` (f32 (N,C,h,w), rows ZERO on entry: the factor loss_weight / count is applied afterwards)."""
    _need_cuda(logits, labels, grad_out)
    logits = _f32c(logits)
    labels = labels.contiguous()
    if labels.dim() == 4:
        labels = labels[:, 0].contiguous()
    if out_size is None:
        out_size = labels.shape[1:3]
    if tuple(labels.shape) != (logits.shape[0], int(out_size[0]), int(out_size[1])):
        raise ValueError('ce: labels shape {} does not match (N,H,W)=({}, {}, {})'.format(
            tuple(labels.shape), logits.shape[0], out_size[0], out_size[1]))
    if grad_out.dtype != torch.float32 or tuple(grad_out.shape) != tuple(logits.shape) or not grad_out.is_contiguous():
        raise ValueError('ce_fused: grad_out must be a contiguous f32 tensor of the logits\' shape')
    d = _ce_desc(logits, labels, ignore_index, out_size, align_corners)
    if not fn['cms_ce_fused_supported'](C.byref(d)):
        sc, cctx = ce_forward(logits, labels, out_size, ignore_index, align_corners, loss_weight, group, sync_count)
        ce_backward(cctx, sc, grad_out)
        return sc
    dev = logits.device
    ws = torch.empty(max(int(fn['cms_ce_workspace_bytes'](C.byref(d))), 16), dtype=torch.uint8, device=dev)
    stats = torch.empty(2, dtype=torch.float64, device=dev)
    check(fn['cms_ce_fwd_bwd'](C.byref(d), _ptr(ws), _ptr(stats), _ptr(grad_out), _stream()), 'cms_ce_fwd_bwd')
    if sync_count:
        import torch.distributed as dist
        cnt = stats[1:2].clone()
        if _allreduce_sum(cnt, group):
            stats = torch.stack([stats[0], cnt[0] / dist.get_world_size(group)])
    scalars = torch.empty(2, dtype=torch.float32, device=dev)
    check(fn['cms_ce_finalize'](_ptr(stats), float(loss_weight), _ptr(scalars), _stream()), 'cms_ce_finalize')
    check(fn['cms_scale_by_scalar'](_ptr(grad_out), grad_out.numel(), _ptr(scalars), 1, 1.0, _stream()), 'cms_scale_by_scalar')
    return scalars


def ce_backward(ctx, scalars, grad_out=None):
    d, keep, _ = ctx
    if grad_out is None:
        grad_out = torch.zeros_like(keep[0])
    check(fn['cms_ce_bwd'](C.byref(d), _ptr(scalars), _ptr(grad_out), _stream()), 'cms_ce_bwd')
    return grad_out


class _CeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, out_size, ignore_index, align_corners):
        scalars, c = ce_forward(logits.detach(), labels, out_size, ignore_index, align_corners)
        ctx.c = c
        ctx.scalars = scalars
        ctx.in_dtype = logits.dtype
        return scalars[0]

    @staticmethod
    def backward(ctx, g):
        sc = ctx.scalars.clone()
        sc[1] = sc[1] * g
        return ce_backward(ctx.c, sc).to(ctx.in_dtype), None, None, None, None


def cross_entropy(logits, labels, out_size=None, ignore_index=255, align_corners=True):
    """nn.CrossEntropyLoss(ignore_index)(upsample(logits), labels) with the upsample fused (pass full-res logits for
    the plain loss)."""
    return _CeFn.apply(logits, labels, None if out_size is None else tuple(out_size), ignore_index, align_corners)


# ---------------------------------------------------------------------------------------------- bilinear upsample
class _UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, align_corners):
        _need_cuda(x)
        xin = _f32c(x)
        n, c, h, w = (int(s) for s in xin.shape)
        H, W = int(size[0]), int(size[1])
        out = torch.empty((n, c, H, W), dtype=torch.float32, device=x.device)
        check(fn['cms_upsample_bilinear_fwd'](_ptr(xin), _ptr(out), n, c, h, w, H, W, int(bool(align_corners)),
                                              _stream()), 'cms_upsample_bilinear_fwd')
        ctx.geo = (n, c, h, w, H, W, int(bool(align_corners)))
        ctx.in_dtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        n, c, h, w, H, W, ac = ctx.geo
        g = _f32c(g)
        gi = torch.empty((n, c, h, w), dtype=torch.float32, device=g.device)
        check(fn['cms_upsample_bilinear_bwd'](_ptr(g), _ptr(gi), n, c, h, w, H, W, ac, _stream()),
              'cms_upsample_bilinear_bwd')
        return gi.to(ctx.in_dtype), None, None


def upsample_bilinear(x, size, align_corners=True):
    return _UpsampleFn.apply(x, tuple(size), align_corners)


# ---------------------------------------------------------------------------------------------- stream pool
_STREAM_POOL = {}


_ROLE_ALIASES = {'1': {'wgrad0': 'teacher', 'optimizer': 'wgrad1'}, 'opt': {'optimizer': 'wgrad1'}, 'tea': {'wgrad0': 'teacher'}, '0': {}}
_GOOD_STREAMS = {}          # device index -> candidate side streams that run CONCURRENTLY with the default stream (probed once)
_STREAM_PROBE_LOG = []


def _probe_side_streams(dev):
    """Side streams that really run beside the current stream AND beside each other. The runtime multiplexes streams onto a few
    hardware queues (4 here); which streams share one depends on what was created / used before (PyTorch's pool, RCCL, a
    notebook's leftovers), and two streams on one queue execute in order: a step whose teacher or weight-gradient stream shares
    a queue with the main stream -- or with each other -- loses 20-45 % (profiles/r05h_*, r05k_*, r05l_*: 482 img/s against 635 with
    ONE foreign stream created first). Probe (once per device and process, a few host synchronisations): every pair out of
    {current stream, six candidates} runs a ~100 us spin kernel side by side; same queue <=> the pair takes twice as long. Result:
    the largest set of candidates that overlap with the current stream and with one another (three, with four queues).
    CMS_STREAM_PROBE=0 switches the probe off (streams are then taken in creation order)."""
    good = _GOOD_STREAMS.get(dev.index)
    if good is not None:
        return good
    n_cand = 6
    cands = [torch.cuda.Stream(device=dev) for _ in range(n_cand)]
    if _os.environ.get('CMS_STREAM_PROBE', '1') == '0' or not hasattr(torch.cuda, '_sleep'):
        good = _GOOD_STREAMS[dev.index] = cands
        return good
    cur = torch.cuda.current_stream(dev)
    spin = 200000                                         # cycles of torch.cuda._sleep: ~0.1 ms
    streams = [cur] + cands

    def pair_ms(a, b):
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(a)
        with torch.cuda.stream(a):
            torch.cuda._sleep(spin)
        if b is not None:
            b.wait_event(e0)
            with torch.cuda.stream(b):
                torch.cuda._sleep(spin)
            a.wait_stream(b)
        e1.record(a)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1)

    with torch.cuda.device(dev):
        for st_ in streams:                               # every stream has run something: its queue is assigned
            with torch.cuda.stream(st_):
                torch.cuda._sleep(1000)
        n = len(streams)
        # The spin kernel counts SHADER clocks, so its duration moves with the power state (0.090 ms at 2.2 GHz, 0.150 ms right after
        # the GPU idled: tests/test_gpu_executor.py met that, round 6). A yardstick taken at a low clock would hide every clash
        # (2 x 0.09 < 1.5 x 0.15): a few ms of work bring the clock up first, the yardstick is measured before AND after the pairs,
        # and a probe whose two yardsticks disagree by more than 20 % is taken again (at most twice).
        for attempt in range(3):
            torch.cuda._sleep(20 * spin)
            pair_ms(cur, cands[0])                        # warm-up
            alone = max(min(pair_ms(cur, None) for _ in range(3)), 1e-3)
            clash = [[False] * n for _ in range(n)]

            def clashes(a, b):
                # noise (another process on the GPU, a second rank on the same device) only ever LENGTHENS a trial: a pair shares a
                # queue only if every one of three trials says so (one 0.1 ms timing per pair marked false clashes on a busy GPU)
                return all(pair_ms(a, b) >= 1.5 * alone for _ in range(3))
            for i in range(n):
                for j in range(i + 1, n):
                    clash[i][j] = clash[j][i] = clashes(streams[i], streams[j])
            alone_end = max(min(pair_ms(cur, None) for _ in range(3)), 1e-3)
            if 0.8 <= alone_end / alone <= 1.25:
                break
    # greedy: candidates in creation order that clash neither with the current stream nor with one already chosen
    chosen = []
    for j in range(1, n):
        if not clash[0][j] and all(not clash[j][k] for k in chosen):
            chosen.append(j)
    _STREAM_PROBE_LOG.append((dev.index, round(alone, 4), [[int(v) for v in row] for row in clash], [j - 1 for j in chosen]))
    good = [streams[j] for j in chosen]
    if len(good) < 3:
        # the step wants three side queues (teacher | two weight-gradient streams); say so instead of degrading silently
        import warnings
        warnings.warn('cutmix-semisup-seg_amd: the stream probe found {} side stream(s) that run beside the current stream and one '
                      'another (3 wanted){}; expect a slower step (roles share hardware queues). Clash matrix: {}'.format(
                          len(good), '' if len(good) >= 2 else ' -- falling back to creation order', _STREAM_PROBE_LOG[-1][2]),
                      RuntimeWarning, stacklevel=2)
    if len(good) < 2:                                     # a runtime this model does not fit: keep the creation order
        good = cands
    _GOOD_STREAMS[dev.index] = good
    return good


def probe_streams(device=None, again=False):
    """Run the side-stream probe NOW (instead of lazily inside the first step / recording) and return the clash log entry.
    `again=True` forgets the previous result AND the pooled role streams: call it after anything that creates streams of its own --
    `torch.distributed.init_process_group` + the first collective (RCCL's internal stream then exists and occupies a hardware queue,
    so the probe steers the step's roles away from it) -- and before the first step records its programs."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    if again:
        _GOOD_STREAMS.pop(dev.index, None)
        for key in [k for k in _STREAM_POOL if k[0] == dev.index]:
            del _STREAM_POOL[key]
    _probe_side_streams(dev)
    return _STREAM_PROBE_LOG[-1] if _STREAM_PROBE_LOG else None


_SIDE_WORK = {}     # device index -> side streams holding weight-gradient launches the main stream has not joined yet
_SIDE_RR = {}       # device index -> round-robin counter over the two weight-gradient streams


_SIDE_SLOT = {}     # (device index, layer key) -> 0 / 1: a layer's weight gradients always go to the SAME side stream


_SIDE_ENABLED = True


def set_side_streams_enabled(on):
    """Switch the layer engines' side streams (weight gradients, ASPP branches) on / off; -> the previous setting. Off while a pass
    is captured into a hipGraph on ONE stream (vat.VATMeanTeacherStep._graphed_grads)."""
    global _SIDE_ENABLED
    prev, _SIDE_ENABLED = _SIDE_ENABLED, bool(on)
    return prev


def side_streams_enabled():
    return _SIDE_ENABLED


def layer_wgrad_stream(device, key=None):
    """(round 6) Stream for ONE weight-gradient launch of the layer engines (backbone_hip._HipConvGeneralFn / _HipClassifierFn: the
    DeepLab v3+ head, the U-Nets), or None = issue it on the current stream. The autograd backward of those networks used to run
    data gradient -> weight gradient -> data gradient ... on ONE stream; the head of DeepLab v3+ alone is 10 ms of such a chain per
    66 ms step with a third of the CUs busy (profiles/r05b_step_timeline_v3plus.txt). Weight gradients feed nothing but the
    optimizer: they alternate over the two pooled weight-gradient streams, each forked from the current stream here (the operands
    are ready on it); `join_side_streams` is called by whatever touches the gradient arena next (optimizer launches, gradient
    clears, the bucketed all-reduce). Off under recording, in the deterministic mode (one shared split-K workspace) and with
    CMS_LAYER_WGRAD_SIDE=0."""
    if _REC is not None or _WGRAD_DETERMINISTIC or not _SIDE_ENABLED or _os.environ.get('CMS_LAYER_WGRAD_SIDE', '1') == '0':
        return None
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    # a LAYER keeps its stream (alternating in first-use order): a step with several backward passes (separate passes: cross
    # entropy, then consistency) adds to the same gradient twice, and the read-modify-write of a padded layer's scratch add is
    # ordered only within one stream
    slot = _SIDE_SLOT.get((idx, key))
    if slot is None:
        n = _SIDE_RR.get(idx, 0)
        _SIDE_RR[idx] = n + 1
        slot = _SIDE_SLOT[(idx, key)] = n & 1
    st = pooled_stream(dev, 'wgrad{}'.format(slot))
    cur = torch.cuda.current_stream(idx)
    if st.cuda_stream == cur.cuda_stream:
        return None
    st.wait_stream(cur)
    pend = _SIDE_WORK.setdefault(idx, [])
    if not pend:
        # first side launch of this backward pass: join when the pass ENDS (autograd runs the callback on the thread that called
        # .backward(), behind the last node), so that whoever reads or updates the gradients afterwards -- this package's fused
        # optimizers, a torch optimizer, a test -- sees them complete on its stream
        try:
            torch.autograd.Variable._execution_engine.queue_callback(lambda i=idx: join_side_streams(torch.device('cuda', i)))
        except RuntimeError:
            pass                         # (not inside a backward pass: the explicit joins -- optimizer, gradient clear -- remain)
    if all(p.cuda_stream != st.cuda_stream for p in pend):
        pend.append(st)
    return st


def join_side_streams(device=None):
    """The current stream waits for every weight-gradient launch `layer_wgrad_stream` put on a side stream since the last join."""
    if not _SIDE_WORK:
        return
    idx = None if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    pend = _SIDE_WORK.pop(idx, None)
    if pend:
        cur = torch.cuda.current_stream(idx)
        for st in pend:
            if st.cuda_stream != cur.cuda_stream:
                cur.wait_stream(st)


def pooled_stream(device, role):
    """A process-wide side stream per (device, role): roles 'teacher', 'wgrad0..2', 'side', 'optimizer'. HIP multiplexes its
    streams onto a handful of hardware queues; a process that keeps creating streams (bench.py runs four workloads, a
    notebook rebuilds its step object) ends up with its 'concurrent' streams on ONE queue and loses the overlap the step is
    built on (configs[1] without --freeze_bn as the fourth workload of one process: 277 img/s against 312 alone,
    profiles/r03also_*). Objects that need a side stream take it from here instead of creating their own.
    Round 5: (i) the streams come from a PROBED set that does not share the default stream's hardware queue (`_probe_side_streams`);
    (ii) roles that are never busy at the same time may share one stream (CMS_STREAM_ALIAS: '1' teacher = first weight-gradient stream and
    optimizer behind the second, 'opt' / 'tea' one of the two, '0' none), so that the step needs fewer queues of its own."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    role = _ROLE_ALIASES.get(_os.environ.get('CMS_STREAM_ALIAS', 'opt'), {}).get(str(role), str(role))
    key = (dev.index, str(role))
    st = _STREAM_POOL.get(key)
    if st is None:
        # CMS_STREAM_PRIO="wgrad=-1+teacher=-1" (experiment, read at creation): -1 = high priority for roles with that prefix
        prio = 0
        for item in _os.environ.get('CMS_STREAM_PRIO', '').split('+'):
            if '=' in item and str(role).startswith(item.split('=')[0]):
                prio = int(item.split('=')[1])
        if prio != 0:
            st = torch.cuda.Stream(device=dev, priority=prio)
        else:
            # slot of a role among the probed streams: roles that run side by side sit on different slots -- teacher | first
            # weight-gradient stream (DeepLab v3+'s single one: 'side') | second weight-gradient stream (+ the optimizer's early launch)
            good = _probe_side_streams(dev)
            slot = {'teacher': 0, 'wgrad0': 1, 'side': 1, 'wgrad1': 2, 'optimizer': 2, 'wgrad2': 0, 'fwd_half_s': 1,
                    'fwd_half_t': 2}.get(str(role))
            if slot is None:
                slot = sum(1 for k in _STREAM_POOL if k[0] == dev.index)
            st = good[slot % len(good)]
        _STREAM_POOL[key] = st
    return st


# ---------------------------------------------------------------------------------------------- NHWC data movement (v3+ head)
def _nhwc_ok(*ts):
    """contiguous NHWC bf16 / fp32 CUDA tensors with channels %% 8 == 0: what csrc/nhwc.hip takes."""
    for t in ts:
        if not (t.is_cuda and t.dim() == 4 and t.is_contiguous() and t.dtype in (torch.bfloat16, torch.float32)
                and t.shape[-1] % 8 == 0):
            return False
    return len({t.dtype for t in ts}) == 1


def _channel_copy(src, src_off, src_pitch, dst, dst_off, dst_pitch, rows, channels, row_div=1):
    es = src.element_size()
    check(fn['cms_channel_copy'](src.data_ptr() + src_off * es, int(src_pitch), dst.data_ptr() + dst_off * es, int(dst_pitch),
                                 int(rows), int(channels), _dtype_code(src), int(row_div), _stream()), 'cms_channel_copy')


class _ConcatChannelsFn(torch.autograd.Function):
    """torch.cat(dim = channels) of NHWC tensors; inputs of shape (N, 1, 1, C) are broadcast over the map (the pooled ASPP
    branch, deeplab3plus.py's F.interpolate of a 1 x 1 map). Backward: slice copies, pixel sums for the broadcast inputs."""

    @staticmethod
    def forward(ctx, *xs):
        # (every input 1 x 1 -- a feature map of one pixel: all of them share that geometry)
        big = next((x for x in xs if x.shape[1] * x.shape[2] > 1 or len(xs) == 1), xs[0])
        n, h, w = (int(v) for v in big.shape[:3])
        ctot = sum(int(x.shape[3]) for x in xs)
        out = torch.empty((n, h, w, ctot), dtype=big.dtype, device=big.device)
        off, meta = 0, []
        for x in xs:
            c = int(x.shape[3])
            bc = tuple(x.shape[1:3]) == (1, 1) and (h, w) != (1, 1)
            _channel_copy(x, 0, c, out, off, ctot, n * h * w, c, row_div=h * w if bc else 1)
            meta.append((off, c, bc))
            off += c
        ctx.meta, ctx.geo = meta, (n, h, w, ctot)
        return out

    @staticmethod
    def backward(ctx, g):
        n, h, w, ctot = ctx.geo
        g = g.contiguous()
        outs = []
        for i, (off, c, bc) in enumerate(ctx.meta):
            if not ctx.needs_input_grad[i]:
                outs.append(None)
            elif bc:
                acc = torch.empty((n, c), dtype=torch.float32, device=g.device)
                check(fn['cms_rows_reduce'](g.data_ptr() + off * g.element_size(), ctot, n, h * w, c, _dtype_code(g), _ptr(acc),
                                            1.0, _stream()), 'cms_rows_reduce')
                outs.append(acc.to(g.dtype).view(n, 1, 1, c))
            else:
                d = torch.empty((n, h, w, c), dtype=g.dtype, device=g.device)
                _channel_copy(g, off, ctot, d, 0, c, n * h * w, c)
                outs.append(d)
        return tuple(outs)


def concat_channels(xs):
    """NHWC tensors (N,H,W,Ci) [or (N,1,1,Ci): broadcast] -> (N,H,W,sum Ci) on csrc/nhwc.hip."""
    xs = [x if x.is_contiguous() else x.contiguous() for x in xs]
    if not _nhwc_ok(*xs):
        raise ValueError('concat_channels: contiguous NHWC bf16 / fp32 CUDA tensors with channels % 8 == 0 required')
    return _ConcatChannelsFn.apply(*xs)


class _UpsampleConcatFn(torch.autograd.Function):
    """cat([low, bilinear_upsample(x -> low's size)], channels) in one buffer (deeplab3plus.py:54-55 of the reference:
    F.interpolate(..., mode='bilinear', align_corners=False) then torch.cat) -- NHWC, the upsample written straight into its
    channel slice; backward = slice copy + the gather-form adjoint of the interpolation."""

    @staticmethod
    def forward(ctx, low, x, align_corners):
        n, H, W, cl = (int(v) for v in low.shape)
        _, h, w, cx = (int(v) for v in x.shape)
        out = torch.empty((n, H, W, cl + cx), dtype=low.dtype, device=low.device)
        _channel_copy(low, 0, cl, out, 0, cl + cx, n * H * W, cl)
        check(fn['cms_upsample_nhwc'](_ptr(x), out.data_ptr() + cl * out.element_size(), cl + cx, n, h, w, H, W, cx,
                                      _dtype_code(x), int(bool(align_corners)), 0, _stream()), 'cms_upsample_nhwc')
        ctx.geo = (n, H, W, cl, h, w, cx, bool(align_corners))
        return out

    @staticmethod
    def backward(ctx, g):
        n, H, W, cl, h, w, cx, align = ctx.geo
        g = g.contiguous()
        dlow = dx = None
        if ctx.needs_input_grad[0]:
            dlow = torch.empty((n, H, W, cl), dtype=g.dtype, device=g.device)
            _channel_copy(g, 0, cl + cx, dlow, 0, cl, n * H * W, cl)
        if ctx.needs_input_grad[1]:
            dx = torch.empty((n, h, w, cx), dtype=g.dtype, device=g.device)
            check(fn['cms_upsample_nhwc'](g.data_ptr() + cl * g.element_size(), _ptr(dx), cl + cx, n, h, w, H, W, cx,
                                          _dtype_code(g), int(align), 1, _stream()), 'cms_upsample_nhwc')
        return dlow, dx, None


def upsample_concat(low, x, align_corners=False):
    low = low if low.is_contiguous() else low.contiguous()
    x = x if x.is_contiguous() else x.contiguous()
    if not _nhwc_ok(low, x) or low.shape[0] != x.shape[0]:
        raise ValueError('upsample_concat: contiguous NHWC bf16 / fp32 CUDA tensors with channels % 8 == 0 required')
    return _UpsampleConcatFn.apply(low, x, align_corners)


class _GlobalAvgPoolFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d(1) on NHWC: (N,H,W,C) -> (N,1,1,C), fp32 accumulation."""

    @staticmethod
    def forward(ctx, x):
        n, h, w, c = (int(v) for v in x.shape)
        acc = torch.empty((n, c), dtype=torch.float32, device=x.device)
        check(fn['cms_rows_reduce'](_ptr(x), c, n, h * w, c, _dtype_code(x), _ptr(acc), 1.0 / float(h * w), _stream()),
              'cms_rows_reduce')
        ctx.geo = (n, h, w, c)
        return acc.to(x.dtype).view(n, 1, 1, c)

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.geo
        row = (g.reshape(n, c).float() * (1.0 / float(h * w))).to(g.dtype).contiguous()
        dx = torch.empty((n, h, w, c), dtype=g.dtype, device=g.device)
        _channel_copy(row, 0, c, dx, 0, c, n * h * w, c, row_div=h * w)
        return dx


def global_avg_pool(x):
    x = x if x.is_contiguous() else x.contiguous()
    if not _nhwc_ok(x):
        raise ValueError('global_avg_pool: contiguous NHWC bf16 / fp32 CUDA tensor with channels % 8 == 0 required')
    return _GlobalAvgPoolFn.apply(x)


class _FanoutFn(torch.autograd.Function):
    """k aliases of one tensor whose gradients are summed by ONE launch (cms_add_n) instead of autograd's k - 1 pairwise adds
    (the ASPP input feeds five branches: four 346 MB bf16 adds per cfg 4 pass)."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k = k
        return tuple(x.view(x.shape) for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        live = [g if g.is_contiguous() else g.contiguous() for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        g0 = live[0]
        if not (g0.is_cuda and g0.dtype in (torch.bfloat16, torch.float32) and g0.numel() % 8 == 0 and len(live) <= 6
                and all(g.dtype == g0.dtype and g.shape == g0.shape for g in live)):
            out = live[0]
            for g in live[1:]:
                out = out + g
            return out, None
        out = torch.empty_like(g0)
        ptrs = (C.c_void_p * len(live))(*[g.data_ptr() for g in live])
        check(fn['cms_add_n'](ptrs, len(live), _ptr(out), g0.numel(), _dtype_code(g0), _stream()), 'cms_add_n')
        return out, None


def fanout(x, k):
    return _FanoutFn.apply(x, int(k))


# ---------------------------------------------------------------------------------------------- EMA
def ema_flat(tgt, src, alpha, tgt_bf16=None):
    """tgt = tgt*alpha + src*(1-alpha) over flat fp32 CUDA buffers, reference rounding (optim_weight_ema.py:21-25)."""
    _need_cuda(tgt, src, tgt_bf16)
    if tgt.dtype != torch.float32 or src.dtype != torch.float32 or not tgt.is_contiguous() or not src.is_contiguous():
        raise TypeError('ema_flat: contiguous float32 buffers required')
    if tgt.numel() != src.numel():
        raise ValueError('ema_flat: size mismatch')
    one_minus_alpha = 1.0 - float(alpha)      # Python double, cast to fp32 at the ABI like the tensor-scalar mul
    check(fn['cms_ema_flat'](_ptr(tgt), _ptr(src), tgt.numel(), float(alpha), one_minus_alpha, _ptr(tgt_bf16),
                             _stream()), 'cms_ema_flat')
    return tgt


# ---------------------------------------------------------------------------------------------- evaluation
def argmax_confusion(logits, labels, num_classes, out_size=None, ignore_index=255, align_corners=True, cm=None,
                     want_pred=False):
    """
    Fused upsample + argmax + confusion matrix (train_seg_semisup_mask_mt.py:510-514, evaluation.py:6-37).
    labels (N,H,W) uint8/int64 or None. Returns (cm int64 (C,C) CUDA [accumulated into `cm` when given], pred|None).
    """
    _need_cuda(logits, labels, cm)
    logits = _f32c(logits)
    n, c, h, w = (int(s) for s in logits.shape)
    if c != num_classes:
        raise ValueError('argmax_confusion: logits have {} classes, expected {}'.format(c, num_classes))
    ldt = _lib.LABEL_U8
    if labels is not None:
        if labels.dim() == 4:
            labels = labels[:, 0]
        labels = labels.contiguous()
        ldt = _lib.LABEL_U8 if labels.dtype == torch.uint8 else _lib.LABEL_I64
        if labels.dtype not in (torch.uint8, torch.int64):
            raise TypeError('labels must be uint8 or int64')
        if out_size is None:
            out_size = labels.shape[1:3]
    if out_size is None:
        out_size = (h, w)
    H, W = int(out_size[0]), int(out_size[1])
    if labels is not None and cm is None:
        cm = torch.zeros((c, c), dtype=torch.int64, device=logits.device)
    pred = torch.empty((n, H, W), dtype=torch.uint8, device=logits.device) if want_pred else None
    check(fn['cms_argmax_confusion'](_ptr(logits), _ptr(labels), ldt, -1 if ignore_index is None else int(ignore_index),
                                     n, c, h, w, H, W, int(bool(align_corners)), _ptr(cm), _ptr(pred), _stream()),
          'cms_argmax_confusion')
    return cm, pred


def confusion(truth, pred, num_classes, ignore_index=None, cm=None):
    """uint8 CUDA maps of equal shape -> int64 (C,C) confusion matrix (row = truth, column = prediction)."""
    _need_cuda(truth, pred, cm)
    truth = truth.contiguous()
    pred = pred.contiguous()
    if truth.dtype != torch.uint8 or pred.dtype != torch.uint8:
        raise TypeError('confusion: uint8 maps required')
    if truth.numel() != pred.numel():
        raise ValueError('confusion: size mismatch')
    if cm is None:
        cm = torch.zeros((num_classes, num_classes), dtype=torch.int64, device=truth.device)
    check(fn['cms_confusion'](_ptr(truth), _ptr(pred), truth.numel(), -1 if ignore_index is None else int(ignore_index),
                              int(num_classes), _ptr(cm), _stream()), 'cms_confusion')
    return cm


# ---------------------------------------------------------------------------------------------- BatchNorm (batch stats)
def _world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


class _BatchNormActFn(torch.autograd.Function):
    """relu(batch_norm(x) (+ res)) on NHWC tensors with batch statistics (csrc/bn.hip); under torch.distributed the
    statistics are all-reduced between the two passes of either direction (SyncBN). `groups`: the batch
    is that many equal runs of samples normalised separately (one launch for what the reference does in separate passes)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, running_mean, running_var, momentum, eps, relu, group, groups):
        n_pix = x.numel() // x.shape[-1]
        c = int(x.shape[-1])
        dev = x.device
        mean, rstd, scale, shift = (torch.empty(groups * c, dtype=torch.float32, device=dev) for _ in range(4))
        ws = bn_workspace(n_pix, c, dev, groups)
        world = _world(group)
        if world > 1:
            # SyncBN, sample groups included (round 4): every group's (sum x, sum x^2) summed over the ranks in ONE all-reduce of
            # [G][2][C] doubles, then finalised group by group, in order (the running statistics move once per group, as in the
            # reference's separate passes); count = the group's pixels on ALL ranks
            stats = torch.empty(groups * 2 * c, dtype=torch.float64, device=dev)
            check(fn['cms_bn_reduce_ws'](_ptr(x), None, None, _dtype_code(x), None, None, _ptr(stats), n_pix, c, groups, 0, _ptr(ws),
                                         _stream()), 'cms_bn_reduce_ws')
            _allreduce_sum(stats, group)
            count = float(n_pix // groups) * world       # (equal shards: the per-GPU batch is fixed under weak scaling)
            for g in range(groups):
                sl = slice(g * c, (g + 1) * c)
                check(fn['cms_bn_finalize'](_ptr(stats[g * 2 * c:(g + 1) * 2 * c]), count, _ptr(gamma), _ptr(beta), float(eps),
                                            float(momentum), _ptr(mean[sl]), _ptr(rstd[sl]), _ptr(scale[sl]), _ptr(shift[sl]),
                                            _ptr(running_mean), _ptr(running_var), c, _stream()), 'cms_bn_finalize')
        else:                                  # statistics and their finalisation in ONE launch
            count = float(n_pix // groups)
            check(fn['cms_bn_stats'](_ptr(x), _dtype_code(x), n_pix, c, groups, _ptr(gamma), _ptr(beta), float(eps),
                                     float(momentum), _ptr(mean), _ptr(rstd), _ptr(scale), _ptr(shift), _ptr(running_mean),
                                     _ptr(running_var), None, None, _ptr(ws), _stream()), 'cms_bn_stats')
        y = torch.empty_like(x)
        check(fn['cms_bn_apply_groups'](_ptr(x), _ptr(res), _ptr(y), _dtype_code(x), _ptr(scale), _ptr(shift), int(bool(relu)),
                                        n_pix, c, groups, _stream()), 'cms_bn_apply')
        ctx.save_for_backward(x, y if relu else None, mean, rstd, gamma, ws)
        ctx.meta = (n_pix, c, count, group, res is not None, gamma is not None and gamma.requires_grad,
                    beta is not None and beta.requires_grad, groups)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd, gamma, ws = ctx.saved_tensors
        n_pix, c, count, group, has_res, want_g, want_b, groups = ctx.meta
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        sums = torch.empty(groups * 2 * c, dtype=torch.float64, device=x.device)
        check(fn['cms_bn_reduce_ws'](_ptr(x), _ptr(dy), _ptr(y), _dtype_code(x), _ptr(mean), _ptr(rstd), _ptr(sums), n_pix, c,
                                     groups, 1, _ptr(ws), _stream()), 'cms_bn_reduce_ws')
        local = sums
        if _world(group) > 1:
            local = sums.clone()                 # parameter gradients stay local (the arena all-reduce sums them)
            _allreduce_sum(sums, group)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        check(fn['cms_bn_bwd_apply_groups'](_ptr(x), _ptr(dy), _ptr(y), _ptr(dx), _ptr(dres), _dtype_code(x), _ptr(mean),
                                            _ptr(rstd), _ptr(gamma), _ptr(sums), count, n_pix, c, groups, _stream()),
              'cms_bn_bwd_apply')
        local = local.view(groups, 2, c).sum(0)    # the groups' passes add into the same parameter gradients
        dgamma = local[1].float() if want_g else None
        dbeta = local[0].float() if want_b else None
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None


class _FrozenBnActFn(torch.autograd.Function):
    """relu(x * scale + shift (+ res)) on contiguous NHWC tensors, scale / shift fp32 [C] WITHOUT gradient (an eval-mode BatchNorm whose
    affine does not train: the teacher of the VAT trainer): one launch forward (cms_bn_apply), one backward (cms_frozen_bn_act_bwd)."""

    @staticmethod
    def forward(ctx, x, scale, shift, res, relu):
        n_pix = x.numel() // x.shape[-1]
        c = int(x.shape[-1])
        y = torch.empty_like(x)
        check(fn['cms_bn_apply_groups_bits'](_ptr(x), _ptr(res) if res is not None else None, _ptr(y), _dtype_code(x), _ptr(scale), _ptr(shift),
                                             int(bool(relu)), n_pix, c, 1, None, _stream()), 'cms_bn_apply')
        ctx.save_for_backward(scale, y if relu else None)
        ctx.geo = (n_pix, c, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        scale, y = ctx.saved_tensors
        n_pix, c, has_res = ctx.geo
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(dy)
        dres = torch.empty_like(dy) if (has_res and ctx.needs_input_grad[3]) else None
        check(fn['cms_frozen_bn_act_bwd'](_ptr(dy), _ptr(y) if y is not None else None, _ptr(dx), _ptr(dres) if dres is not None else None,
                                          _dtype_code(dy), _ptr(scale), n_pix, c, _stream()), 'cms_frozen_bn_act_bwd')
        return dx, None, None, dres, None


def frozen_bn_act(x_nhwc, scale, shift, relu=False, res=None):
    """relu(x * scale + shift (+ res)): an eval-mode BatchNorm (+ residual, + ReLU) with a non-trainable affine on a contiguous NHWC
    bf16 / fp32 tensor (channels % 8 == 0); scale, shift fp32 [C]."""
    _need_cuda(x_nhwc, scale, shift, res)
    if not x_nhwc.is_contiguous() or x_nhwc.shape[-1] % 8 != 0 or x_nhwc.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError('frozen_bn_act: contiguous NHWC bf16 / fp32 tensor with channels % 8 == 0 required')
    if res is not None and (res.shape != x_nhwc.shape or res.dtype != x_nhwc.dtype or not res.is_contiguous()):
        raise ValueError('frozen_bn_act: residual must match the input')
    if scale.dtype != torch.float32 or shift.dtype != torch.float32 or scale.numel() != x_nhwc.shape[-1] or shift.numel() != x_nhwc.shape[-1]:
        raise ValueError('frozen_bn_act: fp32 scale / shift of one value per channel required')
    return _FrozenBnActFn.apply(x_nhwc, scale.contiguous(), shift.contiguous(), res, bool(relu))


def batch_norm_act(x_nhwc, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, relu=False, res=None,
                   group=None, groups=1):
    """nn.BatchNorm2d in training mode (+ residual add, + ReLU) on a contiguous NHWC tensor (channels %% 8 == 0).
    Updates the running statistics in place like the module does (once per sample group, in order)."""
    _need_cuda(x_nhwc, gamma, beta, running_mean, running_var, res)
    if not x_nhwc.is_contiguous() or x_nhwc.shape[-1] % 8 != 0 or x_nhwc.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError('batch_norm_act: contiguous NHWC bf16 / fp32 tensor with channels % 8 == 0 required')
    if res is not None and (res.shape != x_nhwc.shape or res.dtype != x_nhwc.dtype or not res.is_contiguous()):
        raise ValueError('batch_norm_act: residual must match the input')
    groups = int(groups)
    if groups < 1 or x_nhwc.shape[0] % groups != 0:
        raise ValueError('batch_norm_act: {} sample groups do not divide a batch of {}'.format(groups, x_nhwc.shape[0]))
    return _BatchNormActFn.apply(x_nhwc, gamma, beta, res, running_mean, running_var, momentum, eps, relu, group, groups)


def bn_workspace(n_pixels, c, device, groups=1):
    """Zero-filled workspace of one call site of the atomics-free BatchNorm reductions (cms_bn_workspace_bytes): tile counters
    + partial sums; the kernels leave it ready for their next launch, launches on different streams must not share it."""
    n = int(fn['cms_bn_workspace_bytes'](int(n_pixels), int(c), int(groups)))
    if n == 0:
        raise ValueError('bn_workspace: bad geometry ({} pixel rows, {} channels, {} groups)'.format(n_pixels, c, groups))
    ws = torch.empty((n + 3) // 4, dtype=torch.int32, device=device)
    counters = ((int(c) + 63) // 64 * 4 + 255) // 256 * 64      # int32 words: one counter per 64-channel tile, padded to 256 B
    ws[:min(ws.numel(), counters)].zero_()      # only the counters need a defined start; the partial sums are written first
    return ws


_BN_WHAT = {'reduce': 0, 'finalize': 1, 'apply': 2, 'reduce_bwd': 3, 'bwd_apply': 4, 'count': 5, 'stats': 6, 'finalize_tiles': 7, 'sums_tiles': 8}


def bn_op(what, c=0, dtype=None, n_pixels=0, count=0.0, relu=False, eps=1e-5, momentum=0.1, groups=1, tile_rows=0, **t):
    """One launch of the batch-statistics BatchNorm protocol (csrc/bn.hip) on caller-owned buffers -- issued now, or appended
    to the program being recorded (cms_program_add_bn): the executor's batch-statistics passes (backbone_hip.py) are made of
    these. `what`: reduce | finalize | apply | reduce_bwd | bwd_apply | count | stats (= reduce + finalize in one launch) |
    finalize_tiles (statistics from the tile sums the unit's convolution wrote: ws = conv_igemm's stats['tile_sums'], tile_rows) |
    sums_tiles (backward sums from the tile sums of the data-gradient launch that wrote dy: ws, tile_rows -> sums);
    tensors by keyword (x, res, y, dy, dx, dres, sums, gamma, beta, mean, rstd, scale, shift, running_mean, running_var,
    counter, clear_a, clear_b, ws, mask_bits). `mask_bits` (uint8 [pixel rows][c / 8]): 'apply' writes [y > 0] there as bits,
    'reduce_bwd' (with ws) / 'bwd_apply' read them instead of y (1/16 of its bytes; bit-identical results). With `ws` (bn_workspace) the reductions take the atomics-free kernels; `groups` > 1
    (sample groups normalised separately, include/cutmixseg.h) needs them. `count` = pixels of one group."""
    _need_cuda(*t.values())
    d = _lib.BnOp()
    d.what = _BN_WHAT[what]
    d.dtype = _lib.F32 if dtype == torch.float32 else _lib.BF16
    d.c, d.relu, d.groups = int(c), int(bool(relu)), int(groups)
    for k, v in t.items():
        setattr(d, k, None if v is None else v.data_ptr())
    d.count, d.n_pixels = float(count), int(n_pixels)
    d.eps, d.momentum = float(eps), float(momentum)
    d.reserved = int(tile_rows)
    has_ws = t.get('ws') is not None
    if d.groups > 1 and what in ('reduce', 'reduce_bwd') and not has_ws:
        raise ValueError('bn_op: grouped reductions need a workspace')
    if _REC is not None:
        prog = _REC[0]
        idx = fn['cms_program_add_bn'](prog.h, C.byref(d), _rec_stream_index(), prog.group)
        if idx < 0:
            check(idx, 'cms_program_add_bn')
        prog.keep += [v for v in t.values() if v is not None]
        prog.bn_kinds[what] = prog.bn_kinds.get(what, 0) + 1
        return
    g = lambda k: _ptr(t.get(k))
    G = max(1, d.groups)
    if what == 'reduce' and has_ws:
        check(fn['cms_bn_reduce_ws'](g('x'), None, None, d.dtype, None, None, g('sums'), d.n_pixels, d.c, G, 0, g('ws'),
                                     _stream()), 'cms_bn_reduce_ws')
    elif what == 'reduce':
        check(fn['cms_bn_reduce'](g('x'), None, None, d.dtype, None, None, g('sums'), d.n_pixels, d.c, 0, _stream()), 'cms_bn_reduce')
    elif what == 'stats':
        check(fn['cms_bn_stats'](g('x'), d.dtype, d.n_pixels, d.c, G, g('gamma'), g('beta'), d.eps, d.momentum, g('mean'),
                                 g('rstd'), g('scale'), g('shift'), g('running_mean'), g('running_var'), g('counter'), g('sums'),
                                 g('ws'), _stream()), 'cms_bn_stats')
    elif what == 'finalize_tiles':
        check(fn['cms_bn_finalize_tiles'](g('ws'), int(tile_rows), d.n_pixels, d.c, G, g('gamma'), g('beta'), d.eps, d.momentum,
                                          g('mean'), g('rstd'), g('scale'), g('shift'), g('running_mean'), g('running_var'),
                                          g('counter'), _stream()), 'cms_bn_finalize_tiles')
    elif what == 'sums_tiles':
        check(fn['cms_bn_bwd_sums_tiles'](g('ws'), int(tile_rows), d.n_pixels, d.c, G, g('sums'), _stream()), 'cms_bn_bwd_sums_tiles')
    elif what == 'finalize':
        check(fn['cms_bn_finalize_ex'](g('sums'), d.count, g('gamma'), g('beta'), d.eps, d.momentum, g('mean'), g('rstd'),
                                       g('scale'), g('shift'), g('running_mean'), g('running_var'), d.c, g('clear_a'),
                                       g('clear_b'), g('counter'), _stream()), 'cms_bn_finalize_ex')
    elif what == 'apply':
        check(fn['cms_bn_apply_groups_bits'](g('x'), g('res'), g('y'), d.dtype, g('scale'), g('shift'), d.relu, d.n_pixels, d.c, G,
                                             g('mask_bits'), _stream()), 'cms_bn_apply')
    elif what == 'reduce_bwd' and has_ws and t.get('mask_bits') is not None:
        check(fn['cms_bn_reduce_ws_bits'](g('x'), g('dy'), g('mask_bits'), d.dtype, g('mean'), g('rstd'), g('sums'), d.n_pixels, d.c,
                                          G, g('ws'), _stream()), 'cms_bn_reduce_ws_bits')
    elif what == 'reduce_bwd' and has_ws:
        check(fn['cms_bn_reduce_ws'](g('x'), g('dy'), g('y'), d.dtype, g('mean'), g('rstd'), g('sums'), d.n_pixels, d.c, G, 1,
                                     g('ws'), _stream()), 'cms_bn_reduce_ws')
    elif what == 'reduce_bwd':
        check(fn['cms_bn_reduce'](g('x'), g('dy'), g('y'), d.dtype, g('mean'), g('rstd'), g('sums'), d.n_pixels, d.c, 1, _stream()),
              'cms_bn_reduce')
    elif what == 'bwd_apply':
        check(fn['cms_bn_bwd_apply_groups_bits'](g('x'), g('dy'), g('y'), g('mask_bits'), g('dx'), g('dres'), d.dtype, g('mean'),
                                                 g('rstd'), g('gamma'), g('sums'), d.count, d.n_pixels, d.c, G, _stream()),
              'cms_bn_bwd_apply')
    else:
        check(fn['cms_increment_counter'](g('counter'), _stream()), 'cms_increment_counter')


# ---------------------------------------------------------------------------------------------- stem
def stem_out_hw(h, w):
    """-> (ho, wo, hp, wp): sizes after the 7x7/2 convolution and after the ceil-mode 3x3/2 max-pool."""
    v = [C.c_int() for _ in range(4)]
    check(fn['cms_stem_out_hw'](int(h), int(w), *[C.byref(t) for t in v]), 'cms_stem_out_hw')
    return tuple(t.value for t in v)


def stem_pack_weights(w_packed_khkwcoci, out=None):
    """(49, 64, 3) fp32 / bf16 view of conv1.weight in the arena's physical layout -> fp32 (147 + 176, 64): rows
    0..146 = [c,ky,kx][co], the rest = the bf16 (hi, lo) MFMA fragments of the bf16 forward (csrc/stem.hip)."""
    _need_cuda(w_packed_khkwcoci, out)
    if tuple(w_packed_khkwcoci.shape) != (49, 64, 3) or not w_packed_khkwcoci.is_contiguous():
        raise ValueError('stem_pack_weights: contiguous (49, 64, 3) weight view required')
    if out is None:
        # 147 x 64 floats ([tap][co], the VALU kernels and the image gradient) + 2 x 64 x 22 x 8 bf16 behind them: the
        # (hi, lo) MFMA fragments of the bf16 forward = 176 more rows of 64 floats
        out = torch.empty((147 + 176, 64), dtype=torch.float32, device=w_packed_khkwcoci.device)
    if out.dtype != torch.float32 or out.numel() < (147 + 176) * 64 or not out.is_contiguous():
        raise ValueError('stem_pack_weights: `out` must be a contiguous fp32 buffer of at least (147 + 176) x 64 elements')
    check(fn['cms_stem_pack_weights'](_ptr(w_packed_khkwcoci), _dtype_code(w_packed_khkwcoci), _ptr(out), _stream()),
          'cms_stem_pack_weights')
    return out


def stem_forward(x, w147, scale, bias, out_dtype):
    """relu(bn(conv7x7/2(x))) -> (N, ho, wo, 64) NHWC; x (N, 3, H, W) NCHW-contiguous fp32 / bf16."""
    _need_cuda(x, w147, scale, bias)
    if x.dim() != 4 or x.shape[1] != 3 or not x.is_contiguous():
        raise ValueError('stem_forward: contiguous (N, 3, H, W) input required')
    n, _, h, w = (int(v) for v in x.shape)
    ho, wo, _, _ = stem_out_hw(h, w)
    y = torch.empty((n, ho, wo, 64), dtype=out_dtype, device=x.device)
    check(fn['cms_stem_fwd'](_ptr(x), _dtype_code(x), _ptr(y), _dtype_code(y), _ptr(w147), _ptr(scale), _ptr(bias), n, h, w,
                             _stream()), 'cms_stem_fwd')
    return y


def _pool_out(v, ceil_mode):
    if not ceil_mode:
        return (v + 2 - 3) // 2 + 1
    r = (v + 2 - 3 + 1) // 2 + 1            # ATen's ceil-mode output size for kernel 3, stride 2, padding 1
    return r - 1 if (r - 1) * 2 >= v + 1 else r


def maxpool3x3s2_forward(s, ceil_mode=True, out=None):
    """3x3 / 2 / pad 1 max-pool of an NHWC tensor -> (pooled, argmax uint8). `out`: the pooled tensor's buffer (the persistent
    input buffer of a recorded body pass: backbone_hip._StemFn), else a new tensor."""
    _need_cuda(s, out)
    n, hs, ws, c = (int(v) for v in s.shape)
    hp, wp = _pool_out(hs, ceil_mode), _pool_out(ws, ceil_mode)
    if out is not None:
        if tuple(out.shape) != (n, hp, wp, c) or out.dtype != s.dtype or not out.is_contiguous():
            raise ValueError('maxpool3x3s2_forward: `out` must be a contiguous {} tensor of shape {}'.format(s.dtype, (n, hp, wp, c)))
        p = out
    else:
        p = torch.empty((n, hp, wp, c), dtype=s.dtype, device=s.device)
    idx = torch.empty((n, hp, wp, c), dtype=torch.uint8, device=s.device)
    check(fn['cms_maxpool3x3s2_fwd'](_ptr(s), _ptr(p), _ptr(idx), _dtype_code(s), n, hs, ws, c, int(bool(ceil_mode)),
                                     _stream()), 'cms_maxpool3x3s2_fwd')
    return p, idx


def maxpool3x3s2_relu_backward(dp, idx, s, ceil_mode=True):
    """gradient wrt the PRE-ReLU stem output: [s > 0] * max-pool backward."""
    _need_cuda(dp, idx, s)
    dp = dp.contiguous()
    if dp.dtype != s.dtype or dp.shape != idx.shape:
        raise ValueError('maxpool backward: dtype / shape mismatch')
    n, hs, ws, c = (int(v) for v in s.shape)
    ds = torch.empty_like(s)
    check(fn['cms_maxpool3x3s2_relu_bwd'](_ptr(dp), _ptr(idx), _ptr(s), _ptr(ds), _dtype_code(s), n, hs, ws, c,
                                          int(bool(ceil_mode)), _stream()), 'cms_maxpool3x3s2_relu_bwd')
    return ds


def stem_wgrad(x, ds, dw_khkwcoci, scale):
    """dw (49, 64, 3) fp32 view of conv1.weight's gradient in the arena, accumulated into."""
    _need_cuda(x, ds, dw_khkwcoci, scale)
    if tuple(dw_khkwcoci.shape) != (49, 64, 3) or dw_khkwcoci.dtype != torch.float32 or not dw_khkwcoci.is_contiguous():
        raise ValueError('stem_wgrad: contiguous fp32 (49, 64, 3) gradient view required')
    n, _, h, w = (int(v) for v in x.shape)
    ws, nbytes = None, 0
    if deterministic_wgrad():
        nbytes = int(fn['cms_stem_wgrad_workspace_bytes'](_dtype_code(x), _dtype_code(ds), n, h, w))
        key = ('stem', x.device.index, int(torch.cuda.current_stream().cuda_stream))
        ws = _WGRAD_WS.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = _WGRAD_WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    check(fn['cms_stem_wgrad_ws'](_ptr(x), _dtype_code(x), _ptr(ds), _dtype_code(ds), _ptr(dw_khkwcoci), _ptr(scale), n, h, w,
                                  _ptr(ws), nbytes, _stream()), 'cms_stem_wgrad_ws')


def stem_dgrad(ds, w147, scale, x_shape):
    _need_cuda(ds, w147, scale)
    n, _, h, w = (int(v) for v in x_shape)
    dx = torch.empty((n, 3, h, w), dtype=torch.float32, device=ds.device)
    check(fn['cms_stem_dgrad'](_ptr(ds), _dtype_code(ds), _ptr(w147), _ptr(scale), _ptr(dx), n, h, w, _stream()),
          'cms_stem_dgrad')
    return dx


def bn_fold(flat, idx_weight, idx_bias, idx_mean, idx_var, eps, scale, bias):
    """scale = flat[idx_weight] * rsqrt(flat[idx_var] + eps); bias = flat[idx_bias] - flat[idx_mean] * scale -- frozen BatchNorm of
    all layers folded in one launch (csrc/optim.hip: cms_bn_fold; deeplab2.py:92-107)."""
    _need_cuda(flat, idx_weight, idx_bias, idx_mean, idx_var, scale, bias)
    n = int(scale.numel())
    if flat.dtype != torch.float32 or scale.dtype != torch.float32 or bias.dtype != torch.float32 \
            or any(t.dtype != torch.int64 or int(t.numel()) != n or not t.is_contiguous() for t in (idx_weight, idx_bias, idx_mean, idx_var)) \
            or int(bias.numel()) != n or not (scale.is_contiguous() and bias.is_contiguous() and flat.is_contiguous()):
        raise TypeError('bn_fold: fp32 arena / outputs and int64 index tables of the outputs\' length required')
    check(fn['cms_bn_fold'](_ptr(flat), _ptr(idx_weight), _ptr(idx_bias), _ptr(idx_mean), _ptr(idx_var), n, float(eps), _ptr(scale),
                            _ptr(bias), _stream()), 'cms_bn_fold')


# ---------------------------------------------------------------------------------------------- launch programs
HBM_PEAK_BPS, MFMA_PEAK_FLOPS = 8.0e12, 2.5e15       # MI355X_MICROARCH.md: HBM3E, dense bf16 MFMA


class Program(object):
    """A recorded network pass (csrc/program.hip): launch descriptors over persistent buffers, replayed from C++ with
    one call. Record with `with recording(prog, [main_stream, side_stream, ...]):` around ordinary calls of
    conv_igemm / conv_wgrad / memset_zero / stream_wait -- inside the block they append to the program instead of
    launching. Stream indices are positions in the list given to `recording` / `run`."""

    def __init__(self):
        h = C.c_void_p()
        check(fn['cms_program_create'](C.byref(h)), 'cms_program_create')
        self.h = h
        self.keep = []          # every tensor a descriptor points into stays alive as long as the program
        self.marks = []         # (op index, tag): segment boundaries for callers that interleave host work
        self.group = 0          # interleaving key given to the ops recorded next (run_pair)
        self.flops = 0.0        # algorithmic MFMA FLOPs of one replay (convolutions + weight gradients)
        self.conv_launches = 0
        self.conv_bytes = 0.0   # algorithmic HBM bytes of the convolution launches (operands once, output once)
        self.head_bytes = 0.0   # bytes the ASPP head launches MOVE (incl. the fp32 Z planes the GEMM writes and the gather re-reads)
        self.head_bytes_alg = 0.0   # ... their ALGORITHMIC bytes per SURVEY 8(d): the 2048-channel input once + weights + logits
        self.floor_s = 0.0      # sum over the convolution / weight-gradient launches of max(bytes / 8 TB/s, FLOPs / 2.5 PFLOP/s):
                                # the mixed HBM / MFMA roofline of one replay (bench.py: roofline.mixed)
        self.head_launches = 0
        self.by_route = {}      # kernel route (cms_conv_igemm_route / 'wgrad8' / 'wgrad128') -> [launches, algorithmic bytes, FLOPs]
        self.n_streams = 1
        self.bn_kinds = {}      # BatchNorm launches recorded, by kind ('stats', 'finalize_tiles', 'sums_tiles', 'reduce', ...)
        self.host_ops = []      # (op index, stream index, callable): host work between two launches of a replay -- the
                                # all-reduces of SyncBN statistics (recorded with `host_call`); `run` splits around them

    def __del__(self):
        h, self.h = getattr(self, 'h', None), None
        if h is not None:
            try:
                fn['cms_program_destroy'](h)
            except Exception:
                pass

    def size(self):
        return int(fn['cms_program_size'](self.h))

    def mark(self, tag):
        self.marks.append((self.size(), tag))

    @staticmethod
    def _handles(streams):
        arr = (C.c_void_p * len(streams))()
        for i, st in enumerate(streams):
            arr[i] = st.cuda_stream
        return arr

    def _ensure_sync_flags(self, force=False):
        """(round 6) the flag words of the program's cross-stream syncs (csrc/program.hip: a setter + a polling kernel per sync
        instead of an event record + wait): one zeroed int per sync op + the timeout counter, owned by the program, given to the
        native side before the first replay (again if ops were recorded since). OPT-IN (CMS_PROG_FLAG_SYNC=1, or `force`): the
        flag form removes the ~12 us hole per bottleneck from the data-gradient stream and the step gets SLOWER (626.8 / 628.9
        against 634.2 / 633.8 img/s at cfg 2, profiles/r06ae_*: the weight-gradient streams are released earlier and take CUs the
        data-gradient chain's next launch was about to get), so the default replay keeps hipEventRecord / hipStreamWaitEvent."""
        if not force and _os.environ.get('CMS_PROG_FLAG_SYNC', '0') != '1':
            return
        n = int(fn['cms_program_sync_count'](self.h))
        if n > 0 and getattr(self, '_sync_flags_n', 0) != n:
            if getattr(self, 'sync_flags', None) is not None:
                self.keep.append(self.sync_flags)       # (waiters of a replay in flight may still poll the old words)
            self.sync_flags = torch.zeros(n + 1, dtype=torch.int32, device=torch.device('cuda', torch.cuda.current_device()))
            self._sync_flags_n = n
            check(fn['cms_program_set_sync_flags'](self.h, self.sync_flags.data_ptr(), n + 1), 'cms_program_set_sync_flags')

    def sync_timeouts(self):
        """Waiter kernels that gave up (must stay 0; reads the device)."""
        fl = getattr(self, 'sync_flags', None)
        return 0 if fl is None else int(fl[-1].item())

    def run(self, streams, first=0, last=-1):
        if len(streams) < self.n_streams:
            raise ValueError('program recorded on {} streams, {} given'.format(self.n_streams, len(streams)))
        self._ensure_sync_flags()
        if self.host_ops:
            # a host op recorded at index i happens after launches [0, i) and before launch i: it belongs to the range that
            # STARTS at i (segmented replays call run(first, i) and then run(i, ...)), or to the last range when i == size
            size = self.size()
            end = size if (last < 0 or last >= size) else int(last)
            handles = self._handles(streams)
            pos = int(first)
            for idx, si, call in self.host_ops:          # recorded in program order
                if idx < pos or idx > end or (idx == end and end != size):
                    continue
                if idx > pos:
                    check(fn['cms_program_run'](self.h, pos, idx, handles, len(streams)), 'cms_program_run')
                    pos = idx
                with torch.cuda.stream(streams[si]):
                    call()
            if pos < end:
                check(fn['cms_program_run'](self.h, pos, end, handles, len(streams)), 'cms_program_run')
            return
        check(fn['cms_program_run'](self.h, int(first), int(last), self._handles(streams), len(streams)),
              'cms_program_run')

    def set_timing(self, every_k):
        check(fn['cms_program_set_timing'](self.h, int(every_k)), 'cms_program_set_timing')

    def read_timing(self):
        """-> dict(ms, flops, launches, head_ms, head_launches) of the event-bracketed convolution launches since the
        last call (waits for them)."""
        ms, fl, n = C.c_double(), C.c_double(), C.c_long()
        hms, hn = C.c_double(), C.c_long()
        check(fn['cms_program_read_timing'](self.h, C.byref(ms), C.byref(fl), C.byref(n), C.byref(hms), C.byref(hn)),
              'cms_program_read_timing')
        return dict(ms=ms.value, flops=fl.value, launches=n.value, head_ms=hms.value, head_launches=hn.value)


def run_pair(prog_a, streams_a, prog_b, streams_b):
    """Two programs issued interleaved group by group (student on one stream, teacher on another). Programs with host
    operations between their launches (SyncBN all-reduces) cannot be interleaved from the native side: refused, so that a
    caller can never drop the exchanges silently (`_BodyPairFn` issues such passes one after the other)."""
    if prog_a.host_ops or prog_b.host_ops:
        raise RuntimeError('run_pair: a program with host operations (SyncBN exchanges) must be replayed with Program.run')
    prog_a._ensure_sync_flags()
    prog_b._ensure_sync_flags()
    check(fn['cms_program_run_pair'](prog_a.h, Program._handles(streams_a), len(streams_a), prog_b.h,
                                     Program._handles(streams_b), len(streams_b)), 'cms_program_run_pair')


_REC = None      # (Program, [cuda_stream handles]) while recording


class recording(object):
    def __init__(self, prog, streams):
        self.prog, self.streams = prog, [st.cuda_stream for st in streams]
        prog.n_streams = max(prog.n_streams, len(streams))

    def __enter__(self):
        global _REC
        if _REC is not None:
            raise RuntimeError('nested program recording')
        _REC = (self.prog, self.streams)
        return self.prog

    def __exit__(self, *exc):
        global _REC
        _REC = None


def _rec_stream_index(stream=None):
    h = (torch.cuda.current_stream() if stream is None else stream).cuda_stream
    try:
        return _REC[1].index(h)
    except ValueError:
        raise RuntimeError('op recorded on a stream the program was not told about')


def host_call(call):
    """Host work at this point of the current stream: run now, or -- while recording -- at this point of every replay (between
    two native launch calls: Program.run splits around it). Used for the all-reduce of SyncBN statistics, which is issued by
    torch.distributed and cannot be a launch descriptor."""
    if _REC is None:
        call()
        return
    prog = _REC[0]
    prog.host_ops.append((prog.size(), _rec_stream_index(), call))


def memset_zero(t):
    """t.zero_() -- as a program op while recording."""
    _need_cuda(t)
    if _REC is None:
        t.zero_()
        return t
    if not t.is_contiguous():
        raise TypeError('memset_zero: contiguous tensor required')
    prog = _REC[0]
    idx = fn['cms_program_add_memset'](prog.h, _ptr(t), t.numel() * t.element_size(), _rec_stream_index(), prog.group)
    if idx < 0:
        check(idx, 'cms_program_add_memset')
    prog.keep.append(t)
    return t


def stream_wait(waiter, waited):
    """waiter.wait_stream(waited) -- as a program op while recording."""
    if _REC is None:
        waiter.wait_stream(waited)
        return
    prog = _REC[0]
    idx = fn['cms_program_add_sync'](prog.h, _rec_stream_index(waited), _rec_stream_index(waiter), prog.group)
    if idx < 0:
        check(idx, 'cms_program_add_sync')


# ---------------------------------------------------------------------------------------------- ASPP head
def _tap_arrays(taps):
    n = len(taps)
    dy = (C.c_int * n)(*[int(t[0]) for t in taps])
    dx = (C.c_int * n)(*[int(t[1]) for t in taps])
    return dy, dx, n


def aspp_gather_fwd(z, bias, taps, num_classes, out=None):
    """logits[n][c][y][x] = bias[c] + sum_t z[n][t*C + c][y + dy_t][x + dx_t]; z fp32 (N, ZC, h, w) (csrc/aspp.hip)."""
    _need_cuda(z, bias, out)
    if z.dtype != torch.float32 or not z.is_contiguous():
        raise TypeError('aspp_gather_fwd: contiguous fp32 (N, ZC, h, w) input required')
    n, zc, h, w = (int(v) for v in z.shape)
    if out is None:
        out = torch.empty((n, int(num_classes), h, w), dtype=torch.float32, device=z.device)
    dy, dx, nt = _tap_arrays(taps)
    if _REC is not None:
        prog = _REC[0]
        idx = fn['cms_program_add_aspp_gather'](prog.h, _ptr(z), _ptr(bias), _ptr(out), dy, dx, nt, n, int(num_classes), zc,
                                                h, w, _rec_stream_index(), prog.group)
        if idx < 0:
            check(idx, 'cms_program_add_aspp_gather')
        prog.keep += [t for t in (z, bias, out) if t is not None]
        prog.head_bytes += 4.0 * nt * int(num_classes) * n * h * w + 4.0 * out.numel()
        prog.head_bytes_alg += 4.0 * out.numel()
        return out
    check(fn['cms_aspp_gather_fwd'](_ptr(z), _ptr(bias), _ptr(out), dy, dx, nt, n, int(num_classes), zc, h, w, _stream()),
          'cms_aspp_gather_fwd')
    return out


def aspp_spread_bwd(dlogits, taps, zc, dtype, out=None):
    """D[n][y][x][t*C + c] = dlogits[n][c][y - dy_t][x - dx_t] -> NHWC (N, h, w, zc) operand of the head's backward GEMMs."""
    _need_cuda(dlogits, out)
    if dlogits.dtype != torch.float32 or not dlogits.is_contiguous():
        raise TypeError('aspp_spread_bwd: contiguous fp32 (N, C, h, w) gradient required')
    n, c, h, w = (int(v) for v in dlogits.shape)
    if out is None:
        out = torch.empty((n, h, w, int(zc)), dtype=dtype, device=dlogits.device)
    dy, dx, nt = _tap_arrays(taps)
    if _REC is not None:
        prog = _REC[0]
        idx = fn['cms_program_add_aspp_spread'](prog.h, _ptr(dlogits), _ptr(out), _dtype_code(out), dy, dx, nt, n, c,
                                                int(zc), h, w, _rec_stream_index(), prog.group)
        if idx < 0:
            check(idx, 'cms_program_add_aspp_spread')
        prog.keep += [dlogits, out]
        return out
    check(fn['cms_aspp_spread_bwd'](_ptr(dlogits), _ptr(out), _dtype_code(out), dy, dx, nt, n, c, int(zc), h, w, _stream()),
          'cms_aspp_spread_bwd')
    return out


# ---------------------------------------------------------------------------------------------- MFMA convolution
_ZERO_PAGES = {}


def _zero_page(device):
    """An all-zero device buffer: where the direct-to-LDS loader fetches padded / out-of-range rows from (it walks
    through it like through a pixel's channel run, so it must be >= 2 * Cin + 128 bytes: 64 KB covers Cin <= 32704)."""
    z = _ZERO_PAGES.get(device)
    if z is None:
        z = torch.zeros(32768, dtype=torch.bfloat16, device=device)
        _ZERO_PAGES[device] = z
    return z


_CONV8_WS = {}
# cms_conv_igemm_route -> the kernel name rocprofv3 reports (prefix)
_ROUTE_NAMES = {8: 'conv8_kernel', 2: 'conv_igemm_mixed_kernel', 1: 'conv_igemm_kernel<2, 2, 2, 2', 3: 'conv_igemm_kernel<1, 4, 2, 1',
                4: 'conv_igemm_kernel<1, 4, 1, 1', 0: 'other'}


def conv8_workspace(device):
    """Scratch of the eight-phase 256 x 256 convolution (csrc/conv8.hip: arrival counters + stream-K slabs,
    cms_conv_igemm_workspace_bytes). Launches that may overlap must not share it: a recorded program owns one per stream
    it records on (student / teacher passes and the two streams of a backward pass replay concurrently); eager launches
    share one per (device, stream). The counters (first 64 KB) start at zero and every launch leaves them zero."""
    if _REC is not None:
        prog = _REC[0]
        store, key = prog.__dict__.setdefault('conv8_ws', {}), _rec_stream_index()
    else:
        store, key = _CONV8_WS, (device.index, int(torch.cuda.current_stream().cuda_stream))
    ws = store.get(key)
    if ws is None:
        n = int(fn['cms_conv_igemm_workspace_bytes']())
        ws = torch.empty(n, dtype=torch.uint8, device=device)
        ws[:65536].zero_()
        store[key] = ws
    return ws


def conv_taps(kh, kw, dilation, padding):
    """(dy, dx) input offsets of the kh*kw taps in [ky][kx] order."""
    return [(ky * dilation - padding, kx * dilation - padding) for ky in range(kh) for kx in range(kw)]


def conv_igemm(x, w_packed, taps, stride=1, out_hw=None, scale=None, bias=None, res=None, relu=False, mode=0,
               mask_src=None, out=None, out_f32_nchw=None, cout_real=None, out_stride=1, out_full_hw=None, tile=0,
               ksplit=1, variant=0, out_pixel_offset=0, mask_bits_out=None, mask_bits=None, stats=None,
               mask_gates_res=False):
    """
    Implicit-GEMM convolution on the MFMA units (csrc/conv.hip).
      x         bf16 NHWC-contiguous tensor of logical shape (N, H, W, Cin)
      w_packed  bf16 (ntaps, Cout, Cin)
      taps      list of (dy, dx) input offsets per tap
    Returns the bf16 (N, out_h, out_w, Cout) output (or the fp32 NCHW tensor given in `out_f32_nchw`).
    `out_pixel_offset` (with `out`, `out_stride` > 1): the strided scatter starts at that pixel of the output tensor
    instead of pixel 0 -- residual / mask are read at the same shifted positions (the phases of a transposed convolution).
    `mask_bits_out` (forward + ReLU): uint8 (N, out_h, out_w, Cout / 8) that receives [y > 0] as bits; `mask_bits` (mode 1):
    such a tensor INSTEAD of `mask_src` -- the ReLU mask of a data gradient at 1/16 of the bytes (cms_conv_desc).
    `mask_gates_res` (mode 1 with `res` and `mask_bits`): the bits gate the residual only, y = acc + (bit ? res : 0).
    `stats` (bf16 forward launches): a dict {'groups': G}; when the kernel that takes the launch can, its epilogue also writes the
    per-channel (sum, sum of squares) of every pixel tile it stores (cms_conv_desc.stats_out) and the dict comes back with
    'tile_rows' (128 / 256) and 'tile_sums' (fp32 [tiles][2][2][Cout]) for bn_op('finalize_tiles'); 'tile_rows' = 0 when it cannot
    (the statistics then take a pass over the output: bn_op('stats')). Data gradients (mode 1): the dict also carries 'u', 'mean',
    'rstd' and optionally 'bits' of the batch-statistics unit whose OUTPUT gradient the launch writes; the tile sums are then
    (sum d, sum d * xhat) for bn_op('sums_tiles') (cms_conv_desc.bstats_*).
    """
    _need_cuda(x, w_packed, scale, bias, res, mask_src, out, out_f32_nchw, mask_bits_out, mask_bits)
    for t in (mask_bits_out, mask_bits):
        if t is not None and (t.dtype != torch.uint8 or not t.is_contiguous() or x.dtype != torch.bfloat16):
            raise TypeError('conv_igemm: ReLU mask bits are contiguous uint8 tensors of the bf16 entry point')
    if mask_bits is not None and mask_src is not None:
        raise ValueError('conv_igemm: mask_bits replaces mask_src')
    if mask_gates_res and (mask_bits is None or res is None or int(mode) != 1):
        raise ValueError('conv_igemm: mask_gates_res needs a data-gradient launch with res and mask_bits')
    if x.dtype not in (torch.bfloat16, torch.float32) or w_packed.dtype != x.dtype or not x.is_contiguous() \
            or not w_packed.is_contiguous():
        raise TypeError('conv_igemm: contiguous NHWC input and packed weights of one dtype (bf16 or fp32) required')
    for t in (res, mask_src, out):
        if t is not None and (t.dtype != x.dtype or not t.is_contiguous()):
            raise TypeError('conv_igemm: residual / mask / output tensors must be contiguous and of the input dtype')
    f32 = x.dtype == torch.float32          # fp32 parity configuration: csrc/conv_f32.hip
    n, h, w_in, cin = (int(s) for s in x.shape)
    ntaps, cout, cin_w = (int(s) for s in w_packed.shape)
    if cin_w != cin or ntaps != len(taps):
        raise ValueError('conv_igemm: weight / tap table does not match the input')
    ho, wo = (int(out_hw[0]), int(out_hw[1])) if out_hw is not None else (h, w_in)
    oh, ow = (int(out_full_hw[0]), int(out_full_hw[1])) if out_full_hw is not None else (ho, wo)
    d = _lib.ConvDesc()
    d.x, d.w = x.data_ptr(), w_packed.data_ptr()
    if out_f32_nchw is None:
        if out is None:
            out = torch.empty((n, oh, ow, cout), dtype=x.dtype, device=x.device)
            if out_stride > 1:
                memset_zero(out)               # the strided scatter only visits every out_stride-th position
        shift = int(out_pixel_offset) * cout * x.element_size()
        if shift and (out_stride <= 1 or out is None):
            raise ValueError('conv_igemm: out_pixel_offset needs a caller-owned `out` and out_stride > 1')
        d.y, d.y32 = out.data_ptr() + shift, None
    else:
        d.y, d.y32 = None, out_f32_nchw.data_ptr()
    d.scale = scale.data_ptr() if scale is not None else None
    d.bias = bias.data_ptr() if bias is not None else None
    shift = int(out_pixel_offset) * cout * x.element_size() if out_f32_nchw is None else 0
    d.res = res.data_ptr() + shift if res is not None else None
    d.mask_src = mask_src.data_ptr() + shift if mask_src is not None else None
    d.n, d.h, d.w_in, d.cin = n, h, w_in, cin
    bshift = int(out_pixel_offset) * (int(w_packed.shape[1]) // 8) if out_f32_nchw is None else 0
    d.mask_bits_out = mask_bits_out.data_ptr() + bshift if mask_bits_out is not None else None
    d.mask_bits = mask_bits.data_ptr() + bshift if mask_bits is not None else None
    d.mask_gates_res = int(bool(mask_gates_res))
    d.ho, d.wo, d.cout = ho, wo, cout
    d.cout_real = cout if cout_real is None else int(cout_real)
    d.ntaps = ntaps
    for i, (dy, dx) in enumerate(taps):
        d.tap_dy[i], d.tap_dx[i] = int(dy), int(dx)
    d.stride = int(stride)
    d.out_h, d.out_w, d.out_stride = oh, ow, int(out_stride)
    d.relu, d.mode, d.tile = int(bool(relu)), int(mode), int(tile)
    d.ksplit = int(ksplit)
    zp = _zero_page(x.device)
    d.zeros, d.zeros_bytes = zp.data_ptr(), zp.numel() * 2
    d.variant = int(variant)
    if not f32 and out_f32_nchw is None and cout % 256 == 0 and cin % 64 == 0 and ksplit <= 1 \
            and ((int(variant) == 0 and ntaps * (cin // 64) >= 8) or int(variant) in (91, 93)):
        ws = conv8_workspace(x.device)       # wide, K-deep layer: the library may take the eight-phase kernel (csrc/conv8.hip)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    tile_sums = None
    if stats is not None:
        stats['tile_rows'], stats['tile_sums'] = 0, None
        G = max(1, int(stats.get('groups', 1)))
        M = n * ho * wo
        if not f32 and out_f32_nchw is None and M % G == 0:
            d.stats_rows_per_group = M // G
            if stats.get('u') is not None:          # backward statistics of the unit whose output gradient this launch writes
                u_, mean_, rstd_, bits_ = stats['u'], stats['mean'], stats['rstd'], stats.get('bits')
                _need_cuda(u_, mean_, rstd_, bits_)
                if u_.dtype != x.dtype or not u_.is_contiguous() or u_.numel() != M * cout or mean_.numel() != G * cout \
                        or rstd_.numel() != G * cout or (bits_ is not None and (bits_.dtype != torch.uint8 or bits_.numel() * 8 != M * cout)):
                    raise ValueError('conv_igemm: backward statistics need u like the output, mean / rstd [groups][Cout], bits [pixels][Cout / 8]')
                d.bstats_u, d.bstats_mean, d.bstats_rstd = u_.data_ptr(), mean_.data_ptr(), rstd_.data_ptr()
                d.bstats_bits = bits_.data_ptr() if bits_ is not None else None
            rows = int(fn['cms_conv_igemm_stats_tile_rows'](C.byref(d)))
            if rows < 0:
                check(rows, 'cms_conv_igemm_stats_tile_rows')
            if rows > 0:
                tile_sums = torch.empty(((M + rows - 1) // rows) * 4 * cout, dtype=torch.float32, device=x.device)
                d.stats_out = tile_sums.data_ptr()
                stats['tile_rows'], stats['tile_sums'] = rows, tile_sums
    if _REC is not None:
        prog = _REC[0]
        idx = fn['cms_program_add_conv'](prog.h, C.byref(d), int(f32), _rec_stream_index(), prog.group)
        if idx < 0:
            check(idx, 'cms_program_add_conv')
        prog.keep += [t for t in (x, w_packed, out, out_f32_nchw, scale, bias, res, mask_src, zp, mask_bits_out, mask_bits, tile_sums)
                      if t is not None]
        if stats is not None:
            prog.keep += [t for t in (stats.get('u'), stats.get('mean'), stats.get('rstd'), stats.get('bits')) if t is not None]
        esz = x.element_size()
        prog.flops += 2.0 * n * ho * wo * cout * cin * ntaps
        nbytes = esz * (x.numel() + w_packed.numel()) + float(n * ho * wo) * (
            (4.0 * d.cout_real if out_f32_nchw is not None else esz * cout)
            + (esz * cout if res is not None else 0.0) + (esz * cout if mask_src is not None else 0.0)
            + (cout / 8.0 if mask_bits is not None else 0.0) + (cout / 8.0 if mask_bits_out is not None else 0.0)
            + ((esz * cout + (cout / 8.0 if stats.get('bits') is not None else 0.0))
               if (stats is not None and stats.get('u') is not None and stats.get('tile_rows')) else 0.0))
        # (the head's fp32 Z planes are an intermediate of the single-pass formulation, not algorithmic output: its floor counts
        # the input and the weights here and the logits at the gather)
        fbytes = esz * (x.numel() + w_packed.numel()) if (out_f32_nchw is not None and d.cout_real > 32) else nbytes
        prog.floor_s += max(fbytes / HBM_PEAK_BPS, 2.0 * n * ho * wo * cout * cin * ntaps / MFMA_PEAK_FLOPS)
        if not f32:
            route = _ROUTE_NAMES.get(int(fn['cms_conv_igemm_route'](C.byref(d))), 'other')
            r = prog.by_route.setdefault(route, [0, 0.0, 0.0])
            r[0] += 1; r[1] += nbytes; r[2] += 2.0 * n * ho * wo * cout * cin * ntaps
        if out_f32_nchw is not None:
            prog.head_launches += 1
            prog.head_bytes += nbytes
            prog.head_bytes_alg += esz * (x.numel() + w_packed.numel())      # (its Z planes are an intermediate, not an output)
        else:
            prog.conv_launches += 1
            prog.conv_bytes += nbytes
        return out if out_f32_nchw is None else out_f32_nchw
    name = 'cms_conv_igemm_f32' if f32 else 'cms_conv_igemm'
    check(fn[name](C.byref(d), _stream()), name)
    return out if out_f32_nchw is None else out_f32_nchw


def conv_pack_transpose(w_packed, scale=None, flip=True, out=None, out_dtype=None):
    """(ntaps, Cout, Cin) fp32/bf16 -> bf16 (ntaps, Cin, Cout) with BN scale folded and taps flipped: dgrad operand.
    `out_dtype=torch.float32` (fp32 source only) gives the operand of the fp32 parity configuration."""
    _need_cuda(w_packed, scale, out)
    ntaps, cout, cin = (int(s) for s in w_packed.shape)
    if not w_packed.is_contiguous():
        raise TypeError('conv_pack_transpose: contiguous weights required')
    if out_dtype == torch.float32 or (out is not None and out.dtype == torch.float32):
        if w_packed.dtype != torch.float32:
            raise TypeError('conv_pack_transpose: an fp32 destination needs an fp32 source')
        if out is None:
            out = torch.empty((ntaps, cin, cout), dtype=torch.float32, device=w_packed.device)
        check(fn['cms_conv_pack_transpose_f32'](_ptr(w_packed), _ptr(out), _ptr(scale), ntaps, cout, cin,
                                                int(bool(flip)), _stream()), 'cms_conv_pack_transpose_f32')
        return out
    if out is None:
        out = torch.empty((ntaps, cin, cout), dtype=torch.bfloat16, device=w_packed.device)
    check(fn['cms_conv_pack_transpose'](_ptr(w_packed), _dtype_code(w_packed), _ptr(out), _ptr(scale), ntaps, cout, cin,
                                        int(bool(flip)), _stream()), 'cms_conv_pack_transpose')
    return out


class PackTransposePlan(object):
    """Device-resident item table for cms_conv_pack_transpose_batch: (src (ntaps,Cout,Cin), dst, scale) triples with
    fixed addresses, re-packed in ONE launch (`run()`)."""

    def __init__(self, triples):
        items = (_lib.PackItem * len(triples))()
        blk = 0
        self._keep = triples
        dt = triples[0][0].dtype
        ddt = triples[0][1].dtype
        if ddt not in (torch.bfloat16, torch.float32) or (ddt == torch.float32 and dt != torch.float32):
            raise TypeError('PackTransposePlan: bf16 destinations, or fp32 destinations from fp32 sources')
        self.dst_f32 = ddt == torch.float32
        # (round 6) 64 x 64 tiles with 16-byte accesses where every tensor allows it (bf16 -> bf16, channel counts % 64 == 0: the
        # DeepLab bodies); CMS_PACK64=0 keeps the 32 x 32 kernel (A/B)
        self.tile64 = (dt == torch.bfloat16 and ddt == torch.bfloat16 and _os.environ.get('CMS_PACK64', '1') != '0'
                       and all(int(s.shape[1]) % 64 == 0 and int(s.shape[2]) % 64 == 0 for s, _, _ in triples))
        tl = 64 if self.tile64 else 32
        for i, (src, dst, scale) in enumerate(triples):
            _need_cuda(src, dst, scale)
            ntaps, cout, cin = (int(v) for v in src.shape)
            if src.dtype != dt or not src.is_contiguous() or not dst.is_contiguous() \
                    or tuple(dst.shape) != (ntaps, cin, cout) or dst.dtype != ddt:
                raise TypeError('PackTransposePlan: contiguous (ntaps,Cout,Cin) sources of one dtype, '
                                '(ntaps,Cin,Cout) destinations of one dtype')
            items[i].src, items[i].dst = src.data_ptr(), dst.data_ptr()
            items[i].scale = scale.data_ptr() if scale is not None else None
            items[i].ntaps, items[i].cout, items[i].cin, items[i].first_block = ntaps, cout, cin, blk
            blk += ntaps * ((cout + tl - 1) // tl) * ((cin + tl - 1) // tl)
        self.n_items, self.total_blocks = len(triples), blk
        self.dtype_code = _dtype_code(triples[0][0])
        self.table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(triples[0][0].device)

    def run(self):
        if self.dst_f32:
            check(fn['cms_conv_pack_transpose_batch_f32'](_ptr(self.table), self.n_items, self.total_blocks, _stream()),
                  'cms_conv_pack_transpose_batch_f32')
            return
        if self.tile64:
            check(fn['cms_conv_pack_transpose_batch64'](_ptr(self.table), self.n_items, self.total_blocks, _stream()),
                  'cms_conv_pack_transpose_batch64')
            return
        check(fn['cms_conv_pack_transpose_batch'](_ptr(self.table), self.n_items, self.total_blocks, self.dtype_code,
                                                  _stream()), 'cms_conv_pack_transpose_batch')


# Deterministic weight gradients (csrc/conv.hip: split-K partial sums through slabs + an ordered reduce instead of fp32
# atomics). Off by default: inside the two-stream step the atomics are 1.5-2 % faster (profiles/r03a_*). Switch on with
# `set_deterministic_wgrad(True)` (StepConfig.deterministic, the trainers' --deterministic) or CMS_WGRAD_SLAB=1.
import os as _os
_WGRAD_DETERMINISTIC = _os.environ.get('CMS_WGRAD_SLAB', '0') not in ('', '0', '2')
# CMS_WGRAD_SLAB=2 (experiment): slabs only for the large layers (|dW| >= 256 k elements: layer3 / layer4), atomics for the rest
_WGRAD_SLAB_LARGE = _os.environ.get('CMS_WGRAD_SLAB', '0') == '2'
_WGRAD_WS = {}            # (device index, stream handle) -> uint8 scratch of eagerly issued launches
if _WGRAD_DETERMINISTIC:
    fn['cms_loss_set_deterministic'](1)


def set_deterministic_wgrad(on):
    """Run-to-run deterministic weight gradients for every launch issued (or RECORDED) from now on."""
    global _WGRAD_DETERMINISTIC
    _WGRAD_DETERMINISTIC = bool(on)
    fn['cms_loss_set_deterministic'](int(bool(on)))     # the loss kernels' backward: colour classes instead of one launch


def deterministic_wgrad():
    return _WGRAD_DETERMINISTIC


def _wgrad_workspace(d, device, recording):
    """Scratch of exactly the size this launch needs (cms_conv_wgrad_workspace_bytes). A recorded launch owns its slab
    (kept alive by the program; replays on any stream never share it); eager launches reuse one buffer per stream
    (launches on one stream are serialised), grown on demand."""
    need = int(fn['cms_conv_wgrad_workspace_bytes'](C.byref(d)))
    if need <= 0:
        return None
    if recording:
        return torch.empty(need, dtype=torch.uint8, device=device)
    key = (device.index, int(torch.cuda.current_stream().cuda_stream))
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _WGRAD_WS[key] = torch.empty(need, dtype=torch.uint8, device=device)
    return ws


def conv_wgrad(du, x, taps, dw, stride=1, scale=None, cout_real=None, ksplit=0, w_bf16=None, wdot=None, dbeta=None,
               dw_cout=None, query_kernel=False, wg_target=0):
    """
    dw (fp32 (ntaps, Cout, Cin), accumulated into) += scale[co] * sum_pixels du[pix][co] * x[pix + tap][ci].
    du bf16 (N, Ho, Wo, Cout), x bf16 (N, H, W, Cin), both NHWC-contiguous. `dw_cout`: rows per tap of `dw` when it is
    narrower than du's (padded) channel axis -- only the first `cout_real` rows are written.
    `wg_target`: CUs the eight-phase kernel should aim at when the launch runs BESIDE other work (0 = it has the machine).
    `query_kernel`: launch nothing, return 8 if this call would take the eight-phase 256 x 256 kernel (csrc/wgrad8.hip), else 0.
    """
    _need_cuda(du, x, dw, scale)
    if du.dtype not in (torch.bfloat16, torch.float32) or x.dtype != du.dtype or dw.dtype != torch.float32:
        raise TypeError('conv_wgrad: bf16 (or fp32, parity configuration) activations of one dtype and an fp32 '
                        'gradient buffer required')
    f32 = du.dtype == torch.float32
    if not (du.is_contiguous() and x.is_contiguous() and dw.is_contiguous()):
        raise TypeError('conv_wgrad: contiguous tensors required')
    n, ho, wo, cout = (int(s) for s in du.shape)
    n2, h, w_in, cin = (int(s) for s in x.shape)
    if n2 != n or tuple(dw.shape) != (len(taps), cout if dw_cout is None else int(dw_cout), cin):
        raise ValueError('conv_wgrad: shape mismatch')
    if dw_cout is not None and (cout_real is None or int(cout_real) > int(dw_cout)):
        raise ValueError('conv_wgrad: dw_cout needs cout_real <= dw_cout')
    d = _lib.WgradDesc()
    d.du, d.x, d.dw = du.data_ptr(), x.data_ptr(), dw.data_ptr()
    d.scale = scale.data_ptr() if scale is not None else None
    d.n, d.h, d.w_in, d.cin, d.ho, d.wo, d.cout = n, h, w_in, cin, ho, wo, cout
    d.cout_real = 0 if cout_real is None else int(cout_real)
    d.ntaps = len(taps)
    for i, (dy, dx) in enumerate(taps):
        d.tap_dy[i], d.tap_dx[i] = int(dy), int(dx)
    d.stride = int(stride)
    d.ksplit = int(ksplit)
    # side outputs for a trainable BatchNorm affine behind this convolution (see cms_wgrad_desc)
    _need_cuda(w_bf16, wdot, dbeta)
    d.w = w_bf16.data_ptr() if w_bf16 is not None else None
    d.wdot = wdot.data_ptr() if wdot is not None else None
    d.dbeta = dbeta.data_ptr() if dbeta is not None else None
    d.dw_cout = 0 if dw_cout is None else int(dw_cout)
    d.wg_target = int(wg_target)
    if query_kernel:
        return 8 if (not f32 and int(fn['cms_conv_wgrad_uses_wgrad8'](C.byref(d)))) else 0
    ws = None
    if not f32 and (_WGRAD_DETERMINISTIC or (_WGRAD_SLAB_LARGE and len(taps) * cout * cin >= 262144)):
        ws = _wgrad_workspace(d, du.device, _REC is not None)
        if ws is not None:
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    if _REC is not None:
        prog = _REC[0]
        idx = fn['cms_program_add_wgrad'](prog.h, C.byref(d), int(f32), _rec_stream_index(), prog.group)
        if idx < 0:
            check(idx, 'cms_program_add_wgrad')
        prog.keep += [t for t in (du, x, dw, scale, w_bf16, wdot, dbeta, ws) if t is not None]
        prog.flops += 2.0 * n * ho * wo * cout * cin * len(taps)
        prog.floor_s += max((du.element_size() * (du.numel() + x.numel()) + 4.0 * dw.numel()) / HBM_PEAK_BPS,
                            2.0 * n * ho * wo * cout * cin * len(taps) / MFMA_PEAK_FLOPS)
        if not f32:
            route = 'wgrad8_kernel' if int(fn['cms_conv_wgrad_uses_wgrad8'](C.byref(d))) else 'conv_wgrad_kernel'
            r = prog.by_route.setdefault(route, [0, 0.0, 0.0])
            r[0] += 1; r[1] += du.element_size() * (du.numel() + x.numel()) + 4.0 * dw.numel()
            r[2] += 2.0 * n * ho * wo * cout * cin * len(taps)
        return dw
    name = 'cms_conv_wgrad_f32' if f32 else 'cms_conv_wgrad'
    check(fn[name](C.byref(d), _stream()), name)
    return dw


def channel_sum(src, dst):
    """dst (fp32 [C], accumulated with atomics) += sum over all leading dimensions of src[..., C] (bf16 / fp32, contiguous,
    C % 64 == 0): d(beta) = sum_p dU of a trainable BatchNorm affine over frozen statistics (csrc/wfinish.hip). A program op while
    recording."""
    _need_cuda(src, dst)
    c = int(src.shape[-1])
    if not src.is_contiguous() or src.dtype not in (torch.bfloat16, torch.float32) or dst.dtype != torch.float32 \
            or not dst.is_contiguous() or int(dst.numel()) != c or c % 64:
        raise TypeError('channel_sum: contiguous bf16 / fp32 (..., C) input with C % 64 == 0 and an fp32 [C] output required')
    rows = src.numel() // c
    if _REC is not None:
        prog = _REC[0]
        idx = fn['cms_program_add_channel_sum'](prog.h, _ptr(src), _dtype_code(src), rows, c, _ptr(dst), _rec_stream_index(), prog.group)
        if idx < 0:
            check(idx, 'cms_program_add_channel_sum')
        prog.keep += [src, dst]
        return dst
    check(fn['cms_channel_sum'](_ptr(src), _dtype_code(src), rows, c, _ptr(dst), _stream()), 'cms_channel_sum')
    return dst


def wgrad_finish(items):
    """ONE launch behind the weight gradients of a backward pass whose BatchNorm affine trains over frozen statistics
    (csrc/wfinish.hip): `items` = list of (scratch, grad, w_bf16, scale, wdot) with scratch / grad fp32 (ntaps, Cout, Cin) -- the
    unscaled gradient G the eight-phase kernel wrote and the arena's slice --, w_bf16 the weight in the same layout, scale fp32
    [Cout] or None, wdot fp32 [Cout]: grad += scale * G, wdot += <W, G> per output channel, G cleared. A program op while recording
    (the device-resident item table then belongs to the program)."""
    if not items:
        return
    arr = (_lib.WfinishItem * len(items))()
    keep = []
    for i, (scratch, grad, w, scale, wdot) in enumerate(items):
        _need_cuda(scratch, grad, w, scale, wdot)
        nt, co, ci = (int(v) for v in w.shape)
        if not (scratch.dtype == grad.dtype == wdot.dtype == torch.float32 and w.dtype == torch.bfloat16 and scratch.is_contiguous()
                and grad.is_contiguous() and w.is_contiguous() and tuple(scratch.shape) == tuple(grad.shape) == (nt, co, ci)
                and int(wdot.numel()) == co and wdot.is_contiguous() and ci % 4 == 0
                and (scale is None or (scale.dtype == torch.float32 and int(scale.numel()) == co and scale.is_contiguous()))):
            raise TypeError('wgrad_finish: item {}: fp32 scratch / grad (ntaps, Cout, Cin), bf16 weight of that shape, fp32 [Cout] '
                            'scale / wdot, Cin % 4 == 0 required'.format(i))
        m = arr[i]
        m.scratch, m.grad, m.w, m.wdot = scratch.data_ptr(), grad.data_ptr(), w.data_ptr(), wdot.data_ptr()
        m.scale = scale.data_ptr() if scale is not None else None
        m.ntaps, m.cout, m.cin = nt, co, ci
        keep += [t for t in (scratch, grad, w, scale, wdot) if t is not None]
    total = int(fn['cms_wgrad_finish_pack'](arr, len(items)))
    if total < 0:
        check(total, 'cms_wgrad_finish_pack')
    nbytes = C.sizeof(_lib.WfinishItem) * len(items)
    host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    C.memmove(host.data_ptr(), C.addressof(arr), nbytes)
    dev = items[0][0].device
    table = host.to(dev, non_blocking=True)
    if _REC is not None:
        prog = _REC[0]
        idx = fn['cms_program_add_wgrad_finish'](prog.h, C.c_void_p(table.data_ptr()), len(items), total, _rec_stream_index(), prog.group)
        if idx < 0:
            check(idx, 'cms_program_add_wgrad_finish')
        prog.keep += keep + [table, host]
        torch.cuda.current_stream().synchronize()      # (recorded once: the table is in place before any replay, on whatever stream)
        return
    check(fn['cms_wgrad_finish_run'](C.c_void_p(table.data_ptr()), len(items), total, _stream()), 'cms_wgrad_finish_run')
    table.record_stream(torch.cuda.current_stream())


def conv_wgrad_group(jobs, target_workgroups=0):
    """The weight gradients of MANY layers as one grid per kind (cms_conv_wgrad_group_*): `jobs` = list of
    (du, x, taps, dw, stride, scale) as for `conv_wgrad`. Launches that cannot join a group (channel counts that are not
    multiples of 128, fp32, deterministic slabs) are issued one by one. Under recording the device-resident item tables belong to
    the program. Returns the number of grouped launches."""
    if not jobs:
        return 0
    singles, kinds = [], {1: [], 2: []}
    for job in jobs:
        du, x, taps, dw, stride, scale = job
        _need_cuda(du, x, dw, scale)
        ok = du.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dw.dtype == torch.float32 and du.is_contiguous() \
            and x.is_contiguous() and dw.is_contiguous() and not _WGRAD_DETERMINISTIC and not _WGRAD_SLAB_LARGE
        d = None
        if ok:
            n, ho, wo, cout = (int(v) for v in du.shape)
            n2, h, w_in, cin = (int(v) for v in x.shape)
            if n2 != n or tuple(dw.shape) != (len(taps), cout, cin):
                raise ValueError('conv_wgrad_group: shape mismatch')
            d = _lib.WgradDesc()
            d.du, d.x, d.dw = du.data_ptr(), x.data_ptr(), dw.data_ptr()
            d.scale = scale.data_ptr() if scale is not None else None
            d.n, d.h, d.w_in, d.cin, d.ho, d.wo, d.cout = n, h, w_in, cin, ho, wo, cout
            d.ntaps = len(taps)
            for i, (dy, dx) in enumerate(taps):
                d.tap_dy[i], d.tap_dx[i] = int(dy), int(dx)
            d.stride = int(stride)
            kind = int(fn['cms_conv_wgrad_group_kind'](C.byref(d)))
            if kind in kinds:
                kinds[kind].append((d, job))
                continue
        singles.append(job)
    launched = 0
    for kind, items in kinds.items():
        if len(items) == 1:                    # nothing to group with
            singles.append(items[0][1])
            continue
        if not items:
            continue
        n_items = len(items)
        arr = (_lib.WgradDesc * n_items)(*[d for d, _ in items])
        nbytes = int(fn['cms_conv_wgrad_group_bytes'](n_items))
        host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        total = C.c_int(0)
        check(fn['cms_conv_wgrad_group_pack'](arr, n_items, int(target_workgroups), C.c_void_p(host.data_ptr()), nbytes, C.byref(total)),
              'cms_conv_wgrad_group_pack')
        dev = items[0][1][0].device
        table = host.to(dev, non_blocking=True)
        keep = [table, host] + [t for _, job in items for t in (job[0], job[1], job[3], job[5]) if t is not None]
        if _REC is not None:
            prog = _REC[0]
            idx = fn['cms_program_add_wgrad_group'](prog.h, C.c_void_p(table.data_ptr()), n_items, int(total.value), kind,
                                                   _rec_stream_index(), prog.group)
            if idx < 0:
                check(idx, 'cms_program_add_wgrad_group')
            prog.keep += keep
            for d, (du, x, taps, dw, stride, scale) in items:
                fl = 2.0 * d.n * d.ho * d.wo * d.cout * d.cin * d.ntaps
                prog.flops += fl
                prog.floor_s += max((du.element_size() * (du.numel() + x.numel()) + 4.0 * dw.numel()) / HBM_PEAK_BPS,
                                    fl / MFMA_PEAK_FLOPS)
        else:
            check(fn['cms_conv_wgrad_group_run'](C.c_void_p(table.data_ptr()), n_items, int(total.value), kind, _stream()),
                  'cms_conv_wgrad_group_run')
            table.record_stream(torch.cuda.current_stream())
        launched += 1
    for du, x, taps, dw, stride, scale in singles:
        conv_wgrad(du, x, taps, dw, stride=stride, scale=scale)
    return launched
