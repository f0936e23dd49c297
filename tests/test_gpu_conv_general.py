"""
GPU: the general hand-written convolution path of the engine-object networks (backbone_hip._HipConvGeneralFn: channel
padding to 64, tap chunks for kernels of more than 18 taps, strided forward, phase-decomposed strided data gradient) against
torch's fp64 convolution + autograd on the host -- forward, data gradient and weight gradient, in the fp32 parity
configuration (tight) and in bf16 (bf16 tolerance), for every geometry the DeepLab v3+ head and the U-Nets contain
(architectures/deeplab3plus.py:26-101, resunet.py:36-108, denseunet.py:36-143 and their torchvision encoders).
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

CASES = [
    # name, Cin, Cout, k, stride, pad, dil, H, W
    ('stem_7x7_s2_cin3', 3, 64, 7, 2, 3, 1, 45, 61),
    ('densenet_stem_7x7_cout96', 3, 96, 7, 2, 3, 1, 33, 33),
    ('caffe_1x1_s2', 256, 128, 1, 2, 0, 1, 13, 17),
    ('tv_3x3_s2', 128, 128, 3, 2, 1, 1, 14, 19),
    ('tv_down_1x1_s2', 256, 512, 1, 2, 0, 1, 14, 19),
    ('dense_1x1_cin144', 144, 192, 1, 1, 0, 1, 16, 16),
    ('dense_3x3_cout48', 192, 48, 3, 1, 1, 1, 16, 16),
    ('v3p_concat_3x3_cin304', 304, 256, 3, 1, 1, 1, 17, 25),
    ('v3p_project_cout48', 256, 48, 1, 1, 0, 1, 17, 25),
    ('aspp_3x3_d12', 128, 64, 3, 1, 12, 12, 17, 25),
    ('pool_branch_1x1_map', 128, 64, 1, 1, 0, 1, 1, 1),
    ('head_3x3_d6_cout21', 128, 21, 3, 1, 6, 6, 9, 11),
    ('l4_3x3_d4_tiny_map', 512, 512, 3, 1, 4, 4, 7, 9),
    ('l4_1x1_1024_512_tiny_map', 1024, 512, 1, 1, 0, 1, 7, 9),
]


class _Holder(nn.Module):
    def __init__(self, conv):
        super().__init__()
        self.conv = conv


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_general_conv_forward_dgrad_wgrad_vs_fp64(case, dtype):
    from cutmix_semisup_seg_amd.arena import ensure_arena
    from cutmix_semisup_seg_amd.backbone_hip import hip_conv2d
    name, cin, cout, k, stride, pad, dil, H, W = case
    g = torch.Generator().manual_seed(abs(hash(name)) % 997)
    conv = nn.Conv2d(cin, cout, k, stride, pad, dil, bias=False)
    with torch.no_grad():
        conv.weight.copy_((torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(dtype).float())
    N = 2
    x = torch.randn(N, cin, H, W, generator=g).to(dtype).float()
    xd = x.double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    ref = F.conv2d(xd, wd, None, stride, pad, dil)
    dy = torch.randn(ref.shape, generator=g).to(dtype).float()
    ref.backward(dy.double())
    holder = _Holder(conv).to(DEV)
    arena = ensure_arena(holder, with_grad=True, with_bf16=(dtype == torch.bfloat16))
    xg = x.to(DEV).to(dtype).requires_grad_(True)
    y = hip_conv2d(xg, holder.conv, arena, 'conv.weight', dtype)
    assert tuple(y.shape) == tuple(ref.shape) and y.dtype == dtype
    y.backward(dy.to(DEV).to(dtype))
    rel = lambda a, b: float((a.double().cpu() - b).norm() / (b.norm() + 1e-30))
    e = (rel(y.detach(), ref.detach()), rel(xg.grad, xd.grad), rel(holder.conv.weight.grad, wd.grad))
    tol = (2e-6, 2e-6, 5e-6) if dtype == torch.float32 else (6e-3, 6e-3, 2e-3)
    assert all(a <= b for a, b in zip(e, tol)), (name, e)


# ---------------------------------------------------------------------------------------------------------------------------
# Round 6: large weight gradients of the layer engines on the pooled side streams; derived operands cached per weight version
def _big_layer(dtype, cin=320, cout=256, k=3, hw=(65, 65), n=4):
    from cutmix_semisup_seg_amd.arena import ensure_arena
    g = torch.Generator().manual_seed(11)
    conv = nn.Conv2d(cin, cout, k, 1, k // 2, 1, bias=False)
    with torch.no_grad():
        conv.weight.copy_((torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(dtype).float())
    holder = _Holder(conv).to(DEV)
    arena = ensure_arena(holder, with_grad=True, with_bf16=(dtype == torch.bfloat16))
    x = torch.randn(n, cin, hw[0], hw[1], generator=g).to(DEV).to(dtype)
    dy = torch.randn(n, cout, hw[0], hw[1], generator=g).to(DEV).to(dtype)
    return holder, arena, x, dy


def test_layer_engine_weight_gradient_on_a_side_stream_is_complete_when_backward_returns(monkeypatch):
    """`ops.layer_wgrad_stream`: the weight gradient of a LARGE general-path convolution (>= 2 GFLOP) is issued on a pooled side stream
    beside the data gradient; an autograd end-of-backward callback joins it, so `.grad` read on the caller's stream right behind
    `.backward()` is the finished gradient -- equal (fp32 atomics: to 1e-6) to the one computed on the main stream
    (CMS_LAYER_WGRAD_SIDE=0), with nothing left pending; a small layer stays on the main stream."""
    from cutmix_semisup_seg_amd import ops
    from cutmix_semisup_seg_amd.backbone_hip import hip_conv2d
    dtype = torch.bfloat16
    holder, arena, x, dy = _big_layer(dtype)
    seen = []
    orig = ops.layer_wgrad_stream
    monkeypatch.setattr(ops, 'layer_wgrad_stream', lambda *a, **k: (seen.append(orig(*a, **k)) or seen[-1]))

    def grads(side):
        monkeypatch.setenv('CMS_LAYER_WGRAD_SIDE', '1' if side else '0')
        arena.zero_grad()
        xg = x.clone().requires_grad_(True)
        hip_conv2d(xg, holder.conv, arena, 'conv.weight', dtype).backward(dy)
        gw = holder.conv.weight.grad.detach().clone()          # on the caller's stream, right behind backward()
        gx = xg.grad.detach().clone()
        assert not ops._SIDE_WORK, 'the end-of-backward callback joined the side streams'
        torch.cuda.synchronize()
        return gw, gx
    gw1, gx1 = grads(True)
    assert seen and seen[-1] is not None and seen[-1].cuda_stream != torch.cuda.current_stream().cuda_stream
    gw0, gx0 = grads(False)
    assert seen[-1] is None
    assert torch.equal(gx0, gx1)
    assert float((gw1 - gw0).abs().max()) <= 2e-6 * float(gw0.abs().max())
    assert float(gw0.abs().max()) > 0
    # a small layer (dense-block sized) is not worth a stream switch
    n_before = len(seen)
    h2, a2, x2, dy2 = _big_layer(dtype, cin=192, cout=48, k=3, hw=(16, 16), n=2)
    monkeypatch.setenv('CMS_LAYER_WGRAD_SIDE', '1')
    x2g = x2.clone().requires_grad_(True)
    hip_conv2d(x2g, h2.conv, a2, 'conv.weight', dtype).backward(dy2)
    assert len(seen) == n_before, 'small launches stay on the current stream'


def test_padded_operands_are_cached_per_weight_version():
    """`ParamArena.cached`: the zero-padded weight of a layer whose channel counts are not multiples of 64 is built once per weight
    VERSION (fused optimizer / EMA steps and refresh_bf16 bump it) -- the same tensor across calls, a new one with the new values
    after the weights moved."""
    from cutmix_semisup_seg_amd.backbone_hip import hip_conv2d
    dtype = torch.float32
    holder, arena, x, dy = _big_layer(dtype, cin=144, cout=48, k=1, hw=(16, 16), n=2)
    with torch.no_grad():
        y0 = hip_conv2d(x, holder.conv, arena, 'conv.weight', dtype)
        v, t0 = arena.derived[('conv.weight', 'pad', dtype)]
        hip_conv2d(x, holder.conv, arena, 'conv.weight', dtype)
        assert arena.derived[('conv.weight', 'pad', dtype)][1] is t0 and v == arena.version
        holder.conv.weight.mul_(2.0)                 # the weights move behind the arena's back ...
        arena.touch()                                # ... and whoever moved them says so (what the fused optimizers do)
        y1 = hip_conv2d(x, holder.conv, arena, 'conv.weight', dtype)
        assert arena.derived[('conv.weight', 'pad', dtype)][1] is not t0
    assert float((y1 - 2.0 * y0).abs().max()) <= 1e-5 * float(y0.abs().max())


def test_stream_probe_can_be_rerun_and_roles_get_distinct_queues():
    """`ops.probe_streams(again=True)` (what bench.py / the trainers call behind init_process_group + the first collective): forgets
    the pooled role streams, probes again, and the three side roles of the step come from streams that clash neither with the
    current stream nor with one another."""
    from cutmix_semisup_seg_amd import ops
    foreign = [torch.cuda.Stream() for _ in range(2)]             # what a communication library creates
    for st in foreign:
        with torch.cuda.stream(st):
            torch.zeros(1, device=DEV)
    entry = ops.probe_streams(DEV, again=True)
    assert entry is not None and len(entry[3]) >= 2, entry
    roles = [ops.pooled_stream(DEV, r) for r in ('teacher', 'wgrad0', 'wgrad1')]
    cur = torch.cuda.current_stream().cuda_stream
    ids = [r.cuda_stream for r in roles]
    assert cur not in ids
    if len(entry[3]) >= 3:
        assert len(set(ids)) == 3
    del foreign
