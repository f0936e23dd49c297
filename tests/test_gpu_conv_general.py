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
