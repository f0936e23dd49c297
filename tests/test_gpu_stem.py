"""
GPU: the hand-written stem (csrc/stem.hip: 7x7/2 convolution + frozen BatchNorm + ReLU, ceil-mode 3x3/2 max-pool, and
their backward passes) against plain PyTorch on the same values (architectures/deeplab2.py:140-146, 183-186).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def ops():
    from cutmix_semisup_seg_amd import ops as _ops
    return _ops


def _pack49(w):       # (64, 3, 7, 7) -> the arena's physical (49, 64, 3)
    return w.permute(2, 3, 0, 1).reshape(49, 64, 3).contiguous()


@pytest.mark.parametrize('shape', [(2, 33, 47), (1, 321, 321), (3, 64, 40)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_stem_forward_pool_and_backward(ops, shape, dtype):
    N, H, W = shape
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(N, 3, H, W, generator=g).to(dtype).float()
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).to(dtype).float()
    scale, bias = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    # reference in fp64 on the host
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    s_ref = F.relu(F.conv2d(xd, wd, None, 2, 3) * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1))
    p_ref = F.max_pool2d(s_ref, 3, 2, 1, ceil_mode=True)
    ho, wo, hp, wp = ops.stem_out_hw(H, W)
    assert (ho, wo, hp, wp) == (s_ref.shape[2], s_ref.shape[3], p_ref.shape[2], p_ref.shape[3])
    dp = torch.randn(p_ref.shape, generator=g)
    p_ref.backward(dp.double(), retain_graph=True)
    cu = lambda t: t.to(DEV)
    w147 = ops.stem_pack_weights(cu(_pack49(w).to(dtype)))
    s = ops.stem_forward(cu(x.to(dtype)), w147, cu(scale), cu(bias), dtype)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    torch.testing.assert_close(s.float().cpu().permute(0, 3, 1, 2), s_ref.float().detach(), rtol=tol, atol=tol)
    p, idx = ops.maxpool3x3s2_forward(s)
    # the pool is exact on whatever the stem produced
    assert torch.equal(p.float().permute(0, 3, 1, 2), F.max_pool2d(s.float().permute(0, 3, 1, 2), 3, 2, 1, ceil_mode=True))
    if dtype != torch.float32:
        return          # (bf16 ties move the argmax; the backward is checked on the fp32 path, the kernels are shared)
    ds = ops.maxpool3x3s2_relu_backward(cu(dp.permute(0, 2, 3, 1).contiguous()), idx, s)
    # d loss / d (conv * scale + bias) from the reference graph
    ds_ref = torch.autograd.grad(p_ref, s_ref, dp.double(), retain_graph=True)[0] * (s_ref > 0)
    torch.testing.assert_close(ds.cpu().permute(0, 3, 1, 2), ds_ref.float(), rtol=1e-5, atol=1e-6)
    dw = torch.full((49, 64, 3), 0.25, device=DEV)
    ops.stem_wgrad(cu(x), ds, dw, cu(scale))
    want = _pack49(wd.grad.float()) + 0.25
    assert float((dw.cpu() - want).abs().max()) <= 2e-4 * float(want.abs().max()) + 1e-5
    dx = ops.stem_dgrad(ds, w147, cu(scale), x.shape)
    assert float((dx.cpu() - xd.grad.float()).abs().max()) <= 1e-4 * float(xd.grad.abs().max()) + 1e-6


def test_network_runs_without_library_convolutions(ops):
    """fp32 parity configuration end to end on hand-written kernels: stem + body + head vs the CPU oracle, incl. the
    stem's weight gradient (previously the library's) and the image gradient (VAT)."""
    from architectures import deeplab2
    from oracle import deeplab2 as odl
    C, layers = 5, [1, 1, 1, 1]
    st = odl.closed_form_state(C, layers)
    net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    net.load_state_dict(st)
    net = net.to(DEV)
    net.compute_dtype = torch.float32
    net.engine_kind = 'hip'
    net.train(); net.freeze_batchnorm()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 3, 65, 49, generator=g)
    gr = torch.randn(2, C, 9, 7, generator=g)
    xd = x.to(DEV).requires_grad_(True)
    lo = net.forward_lowres(xd)
    lo.backward(gr.to(DEV))
    leaves = {k: v.clone().requires_grad_(True) for k, v in st.items() if v.dtype == torch.float32 and v.dim() == 4}
    st2 = dict(st); st2.update(leaves)
    xr = x.clone().requires_grad_(True)
    want = odl.forward_lowres(xr, st2, layers, frozen=True)
    want.backward(gr)
    torch.testing.assert_close(lo.detach().cpu(), want.detach(), rtol=1e-4, atol=1e-5)
    named = dict(net.named_parameters())
    for k in ('conv1.weight', 'layer1.0.conv1.weight', 'layer4.0.conv2.weight'):
        gw, ww = named[k].grad.cpu(), leaves[k].grad
        assert float((gw - ww).abs().max()) <= 1e-3 * float(ww.abs().max()) + 1e-7, k
    assert float((xd.grad.cpu() - xr.grad).abs().max()) <= 1e-3 * float(xr.grad.abs().max()) + 1e-8


@pytest.mark.parametrize('shape', [(2, 321, 321), (1, 65, 97), (3, 33, 47)])
def test_stem_forward_bf16_on_the_matrix_cores(ops, shape):
    """bf16 forward = stem_fwd_mfma_kernel (K = (c, ky) x 8 kx, weights as a bf16 hi + lo pair, fp32 accumulation):
    within one bf16 rounding of the fp64 reference on the same bf16 inputs and fp32 weights (the weights are NOT
    rounded to bf16: the pair keeps 16 mantissa bits)."""
    N, H, W = shape
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(N, 3, H, W, generator=g).bfloat16()
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1                      # fp32 weights, not representable in bf16
    scale, bias = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    ref = F.relu(F.conv2d(x.double(), w.double(), None, 2, 3) * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1))
    cu = lambda t: t.to(DEV)
    w147 = ops.stem_pack_weights(cu(_pack49(w)))
    s = ops.stem_forward(cu(x), w147, cu(scale), cu(bias), torch.bfloat16)
    got = s.float().cpu().permute(0, 3, 1, 2).double()
    err = (got - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-4).all()), float((err - 2.0 ** -8 * ref.abs()).max())
    # and the fp32 output of the same inputs (VALU kernel) agrees to fp32 accuracy with the reference
    s32 = ops.stem_forward(cu(x.float()), w147, cu(scale), cu(bias), torch.float32)
    torch.testing.assert_close(s32.cpu().permute(0, 3, 1, 2).double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('shape', [(2, 321, 321), (1, 65, 97), (3, 33, 47)])
def test_stem_wgrad_bf16_on_the_matrix_cores(ops, shape):
    """bf16 image x bf16 dS = stem_wgrad_mfma_kernel (im2col tile built in LDS, transpose reads, fp32 accumulation):
    both operands are exact in bf16, so the result equals the fp64 reference up to fp32 summation order; accumulates
    into the gradient buffer like the VALU kernel."""
    N, H, W = shape
    g = torch.Generator().manual_seed(3 * H + W)
    x = torch.randn(N, 3, H, W, generator=g).bfloat16()
    ho, wo, _, _ = ops.stem_out_hw(H, W)
    ds = (torch.randn(N, ho, wo, 64, generator=g) * (torch.rand(N, ho, wo, 64, generator=g) > 0.5)).bfloat16()
    scale = torch.rand(64, generator=g) + 0.5
    wd = torch.zeros(64, 3, 7, 7, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double(), wd, None, 2, 3)
    y.backward(ds.double().permute(0, 3, 1, 2) * scale.double().view(1, -1, 1, 1))
    want = _pack49(wd.grad.float()) + 0.25
    cu = lambda t: t.to(DEV)
    dw = torch.full((49, 64, 3), 0.25, device=DEV)
    ops.stem_wgrad(cu(x), cu(ds), dw, cu(scale))
    err = float((dw.cpu() - want).abs().max())
    assert err <= 2e-5 * float(want.abs().max()) + 1e-5, err
