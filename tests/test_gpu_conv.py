"""
GPU parity of the MFMA implicit-GEMM convolution (csrc/conv.hip, through the C ABI) against the same op in FP64 ON THE HOST
on identical bf16-rounded inputs (F.conv2d + folded-BN affine + residual + ReLU; autograd for the gradients) -- no library
convolution of the device is involved in the main forward / data-gradient / weight-gradient comparisons. What
separates the kernel from that reference is ONE bf16 rounding of the output (half an ulp: 2^-8 relative) plus fp32
accumulation noise, and the tolerances say exactly that; the weight gradient (fp32 output, exact bf16 products) is held to
fp32 summation noise.
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


def _mk(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen, device=DEV) * scale).bfloat16()


def _d(t):             # device tensor -> fp64 on the host
    return t.detach().double().cpu()


BF16_HALF_ULP = 2.0 ** -8


def _pack(w):          # (Cout, Cin, kh, kw) bf16 -> (taps, Cout, Cin)
    co, ci, kh, kw = w.shape
    return w.permute(2, 3, 0, 1).reshape(kh * kw, co, ci).contiguous()


CASES = [
    # name, N, H, W, Cin, Cout, k, stride, dil
    ('l3_conv3_1x1', 2, 41, 41, 256, 1024, 1, 1, 1),
    ('l3_conv2_3x3_d2', 2, 41, 41, 256, 256, 3, 1, 2),
    ('l4_conv2_3x3_d4', 1, 41, 41, 512, 512, 3, 1, 4),
    ('l2_conv1_1x1_s2', 2, 81, 81, 256, 128, 1, 2, 1),
    ('l1_conv1_1x1_c64', 2, 33, 47, 256, 64, 1, 1, 1),
    ('l1_conv2_3x3_c64', 1, 33, 47, 64, 64, 3, 1, 1),
    ('tail_1x1', 1, 7, 9, 64, 128, 1, 1, 1),
]


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize('epi', ['plain', 'bn_relu', 'bn_res_relu'])
def test_conv_forward(ops, case, epi):
    name, N, H, W, Cin, Cout, k, stride, dil = case
    g = torch.Generator(device=DEV).manual_seed(hash(name) % 1000)
    pad = dil * (k - 1) // 2
    x = _mk((N, H, W, Cin), g)
    w = _mk((Cout, Cin, k, k), g, (2.0 / (Cin * k * k)) ** 0.5)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    scale = (torch.rand(Cout, generator=g, device=DEV) + 0.5) if epi != 'plain' else None
    bias = torch.randn(Cout, generator=g, device=DEV) * 0.1 if epi != 'plain' else None
    res = _mk((N, Ho, Wo, Cout), g) if epi == 'bn_res_relu' else None
    relu = epi != 'plain'
    y = ops.conv_igemm(x, _pack(w), ops.conv_taps(k, k, dil, pad), stride=stride, out_hw=(Ho, Wo), scale=scale,
                       bias=bias, res=res, relu=relu)
    ref = F.conv2d(_d(x).permute(0, 3, 1, 2), _d(w), None, stride, pad, dil)
    if scale is not None:
        ref = ref * _d(scale).view(1, -1, 1, 1) + _d(bias).view(1, -1, 1, 1)
    if res is not None:
        ref = ref + _d(res).permute(0, 3, 1, 2)
    if relu:
        ref = F.relu(ref)
    ref = ref.permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    err = (_d(y) - ref).abs()
    tol = 1.02 * BF16_HALF_ULP * ref.abs() + 1e-5 * float(ref.abs().max())
    assert bool((err <= tol).all()), 'max err {} at scale {}'.format(float(err.max()), float(ref.abs().max()))
    assert float(err.mean()) <= 0.5 * BF16_HALF_ULP * float(ref.abs().mean()) + 1e-6


@pytest.mark.parametrize('variant', [0, 1, 4, 5], ids=['direct_to_lds', 'register_staged', 'direct_to_lds_2stage',
                                                    'direct_to_lds_2x32'])
@pytest.mark.parametrize('tile', [0, 128, 1128, 256, 2256, 64, 32])
def test_conv_tile_variants_agree(ops, tile, variant):
    g = torch.Generator(device=DEV).manual_seed(5)
    x = _mk((2, 20, 23, 128), g)
    w = _mk((128, 128, 3, 3), g, 0.03)
    taps = ops.conv_taps(3, 3, 2, 2)
    a = ops.conv_igemm(x, _pack(w), taps, tile=tile, variant=variant)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, 1, 2, 2).permute(0, 2, 3, 1)
    assert float((a.float() - ref).abs().max()) <= 1e-2 * float(ref.abs().max()) + 1e-2


def test_aspp_head_two_dilations_fp32_nchw(ops):
    """conv_d6(x) + conv_d12(x) + bias as ONE 18-tap implicit GEMM with the class axis padded to 32."""
    g = torch.Generator(device=DEV).manual_seed(9)
    N, H, W, Cin, C = 2, 41, 41, 2048, 21
    x = _mk((N, H, W, Cin), g)
    w6 = _mk((C, Cin, 3, 3), g, 0.01)
    w12 = _mk((C, Cin, 3, 3), g, 0.01)
    b = torch.randn(C, generator=g, device=DEV)
    wp = torch.zeros(18, 32, Cin, dtype=torch.bfloat16, device=DEV)
    wp[:9, :C] = _pack(w6)
    wp[9:, :C] = _pack(w12)
    bias = torch.zeros(32, device=DEV)
    bias[:C] = b
    taps = ops.conv_taps(3, 3, 6, 6) + ops.conv_taps(3, 3, 12, 12)
    out = torch.empty(N, C, H, W, device=DEV)
    ops.conv_igemm(x, wp, taps, bias=bias, out_f32_nchw=out, cout_real=C)
    xf = x.float().permute(0, 3, 1, 2)
    ref = F.conv2d(xf, w6.float(), None, 1, 6, 6) + F.conv2d(xf, w12.float(), None, 1, 12, 12) + b.view(1, -1, 1, 1)
    assert float((out - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-3
    for ks in (2, 6, 18):          # taps split over workgroups, partial sums accumulated with atomics
        out2 = torch.zeros(N, C, H, W, device=DEV)
        ops.conv_igemm(x, wp, taps, bias=bias, out_f32_nchw=out2, cout_real=C, ksplit=ks)
        assert float((out2 - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-3


@pytest.mark.parametrize('case', [('1x1', 2, 41, 41, 256, 1024, 1, 1), ('3x3_d2', 2, 41, 41, 256, 256, 3, 2),
                                  ('3x3_d4_tail', 1, 19, 21, 512, 512, 3, 4)], ids=lambda c: c[0])
def test_conv_dgrad_with_relu_mask(ops, case):
    """dX = conv^T(dU * scale) then masked by the producer's ReLU, via the same kernel on packed-transposed weights."""
    name, N, H, W, Cin, Cout, k, dil = case
    g = torch.Generator(device=DEV).manual_seed(11)
    pad = dil * (k - 1) // 2
    x = _mk((N, H, W, Cin), g)                   # saved input activation (post-ReLU of its producer)
    x = torch.relu(x)
    w = _mk((Cout, Cin, k, k), g, (2.0 / (Cin * k * k)) ** 0.5)
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    dU = _mk((N, H, W, Cout), g)                 # gradient wrt the pre-activation of this conv's BN output
    add = _mk((N, H, W, Cin), g)                 # gradient arriving through a residual branch
    wT = ops.conv_pack_transpose(_pack(w), scale=scale, flip=True)      # (taps, Cin, Cout)
    assert wT.shape == (k * k, Cin, Cout)
    dx = ops.conv_igemm(dU, wT, ops.conv_taps(k, k, dil, pad), res=add, mode=1, mask_src=x)
    xr = _d(x).permute(0, 3, 1, 2).requires_grad_(True)
    # (the kernel's operand is bf16(w * scale), rounded once by the pack kernel: the reference differentiates that product)
    ws = (w.float() * scale.view(-1, 1, 1, 1)).bfloat16()
    y = F.conv2d(xr, _d(ws), None, 1, pad, dil)
    y.backward(_d(dU).permute(0, 3, 1, 2))
    ref = (xr.grad + _d(add).permute(0, 3, 1, 2)) * (_d(x).permute(0, 3, 1, 2) > 0)
    ref = ref.permute(0, 2, 3, 1)
    err = (_d(dx) - ref).abs()
    assert bool((err <= 1.02 * BF16_HALF_ULP * ref.abs() + 1e-5 * float(ref.abs().max())).all()), float(err.max())


def test_conv_dgrad_stride2_scatter(ops):
    g = torch.Generator(device=DEV).manual_seed(12)
    N, H, W, Cin, Cout = 2, 81, 81, 256, 128
    w = _mk((Cout, Cin, 1, 1), g, 0.06)
    dU = _mk((N, 41, 41, Cout), g)
    wT = ops.conv_pack_transpose(_pack(w), flip=True)
    dx = ops.conv_igemm(dU, wT, [(0, 0)], mode=1, out_hw=(41, 41), out_stride=2, out_full_hw=(H, W))
    xr = torch.zeros(N, Cin, H, W, device=DEV, requires_grad=True)
    F.conv2d(xr, w.float(), None, 2).backward(dU.float().permute(0, 3, 1, 2))
    ref = xr.grad.permute(0, 2, 3, 1)
    assert float((dx.float() - ref).abs().max()) <= 1e-2 * float(ref.abs().max()) + 1e-3


def test_conv_rejects_bad_shapes(ops):
    x = torch.zeros(1, 4, 4, 48, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(1, 64, 48, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(ValueError, match='multiple of 64'):
        ops.conv_igemm(x, w, [(0, 0)])


WG_CASES = [('1x1_256_1024', 2, 41, 41, 256, 1024, 1, 1, 1), ('3x3d2_256', 2, 41, 41, 256, 256, 3, 1, 2),
            ('1x1s2_256_128', 2, 81, 81, 256, 128, 1, 2, 1), ('3x3_64_64', 2, 33, 47, 64, 64, 3, 1, 1),
            ('1x1_64_256', 1, 33, 47, 64, 256, 1, 1, 1), ('3x3d4_512_tail', 1, 19, 21, 512, 512, 3, 1, 4)]


@pytest.mark.parametrize('case', WG_CASES, ids=[c[0] for c in WG_CASES])
def test_conv_wgrad(ops, case):
    name, N, H, W, Cin, Cout, k, stride, dil = case
    g = torch.Generator(device=DEV).manual_seed(21)
    pad = dil * (k - 1) // 2
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    x = _mk((N, H, W, Cin), g)
    dU = _mk((N, Ho, Wo, Cout), g)
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    dw = torch.zeros(k * k, Cout, Cin, device=DEV)
    ops.conv_wgrad(dU, x, ops.conv_taps(k, k, dil, pad), dw, stride=stride, scale=scale)
    w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(_d(x).permute(0, 3, 1, 2), w, None, stride, pad, dil) * _d(scale).view(1, -1, 1, 1)
    y.backward(_d(dU).permute(0, 3, 1, 2))
    ref = w.grad.permute(2, 3, 0, 1).reshape(k * k, Cout, Cin)
    err = (_d(dw) - ref).abs()
    # fp32 output, exact bf16 x bf16 products: what is left is fp32 summation noise over N * Ho * Wo pixels
    assert float(err.max()) <= 2e-5 * float(ref.abs().max()), (float(err.max()), float(ref.abs().max()))
    # accumulation semantics: a second call adds
    ops.conv_wgrad(dU, x, ops.conv_taps(k, k, dil, pad), dw, stride=stride, scale=scale, ksplit=3)
    assert float((_d(dw) - 2 * ref).abs().max()) <= 4e-5 * float(ref.abs().max())


def test_aspp_wgrad_padded_classes(ops):
    g = torch.Generator(device=DEV).manual_seed(22)
    N, H, W, Cin, C = 2, 41, 41, 2048, 21
    x = _mk((N, H, W, Cin), g)
    dL = torch.zeros(N, H, W, 64, dtype=torch.bfloat16, device=DEV)
    dL[..., :C] = _mk((N, H, W, C), g)
    taps = ops.conv_taps(3, 3, 6, 6) + ops.conv_taps(3, 3, 12, 12)
    dw = torch.zeros(18, 64, Cin, device=DEV)
    ops.conv_wgrad(dL, x, taps, dw, cout_real=C)
    xf = x.float().permute(0, 3, 1, 2)
    for i, d in enumerate((6, 12)):
        w = torch.zeros(C, Cin, 3, 3, device=DEV, requires_grad=True)
        F.conv2d(xf, w, None, 1, d, d).backward(dL[..., :C].float().permute(0, 3, 1, 2))
        ref = w.grad.permute(2, 3, 0, 1).reshape(9, C, Cin)
        got = dw[9 * i:9 * i + 9, :C]
        assert float((got - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-3
    assert float(dw[:, C:].abs().max()) == 0.0


@pytest.mark.parametrize('variant', [0, 1, 5], ids=['direct_to_lds', 'register_staged', 'direct_to_lds_2x32'])
@pytest.mark.parametrize('tile', [0, 1128, 256, 2256, 64, 32])
@pytest.mark.parametrize('epi', ['res', 'mask', 'mask_res', 'fwd_res_relu'])
def test_conv_epilogue_operands_through_lds(ops, epi, tile, variant):
    """Residual / ReLU-mask tiles are staged global -> LDS (swizzled rows) and consumed in accumulator layout; the
    output tile is written in place. Checked on a ragged pixel count (dead rows) for every tile shape; the element
    that must come out is exactly bf16(acc [+ res]) [masked], so the comparison is tight."""
    g = torch.Generator(device=DEV).manual_seed(21)
    N, H, W, Cin, Cout = 3, 13, 17, 64, 256          # 663 pixels: 5 full 128-pixel tiles + a 23-pixel tail
    x = _mk((N, H, W, Cin), g)
    w = _mk((Cout, Cin, 1, 1), g, 0.1)
    res = _mk((N, H, W, Cout), g)
    msk = _mk((N, H, W, Cout), g)
    acc = torch.einsum('nhwc,oc->nhwo', x.float(), w.float()[:, :, 0, 0])
    kw = dict(tile=tile, variant=variant)
    if epi == 'res':
        got, ref = ops.conv_igemm(x, _pack(w), [(0, 0)], mode=1, res=res, **kw), acc + res.float()
    elif epi == 'mask':
        got, ref = ops.conv_igemm(x, _pack(w), [(0, 0)], mode=1, mask_src=msk, **kw), acc * (msk.float() > 0)
    elif epi == 'mask_res':
        got = ops.conv_igemm(x, _pack(w), [(0, 0)], mode=1, mask_src=msk, res=res, **kw)
        ref = (acc + res.float()) * (msk.float() > 0)
    else:
        scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
        bias = torch.randn(Cout, generator=g, device=DEV)
        got = ops.conv_igemm(x, _pack(w), [(0, 0)], scale=scale, bias=bias, res=res, relu=True, **kw)
        ref = torch.relu(acc * scale + bias + res.float())
    err = (got.float() - ref).abs()
    assert bool((err <= 6e-3 * ref.abs() + 2e-3).all()), float(err.max())       # one bf16 rounding + fp32 sum order
    if 'mask' in epi:
        assert bool((got[msk <= 0] == 0).all())


def test_conv_dgrad_stride2_scatter_with_mask_and_residual(ops):
    g = torch.Generator(device=DEV).manual_seed(13)
    N, H, W, Cin, Cout = 2, 21, 23, 256, 128
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    w = _mk((Cout, Cin, 1, 1), g, 0.06)
    dU = _mk((N, Ho, Wo, Cout), g)
    res = _mk((N, H, W, Cin), g)
    xin = _mk((N, H, W, Cin), g)
    wT = ops.conv_pack_transpose(_pack(w), flip=True)
    dx = ops.conv_igemm(dU, wT, [(0, 0)], mode=1, out_hw=(Ho, Wo), out_stride=2, out_full_hw=(H, W), res=res,
                        mask_src=xin)
    xr = torch.zeros(N, Cin, H, W, device=DEV, requires_grad=True)
    F.conv2d(xr, w.float(), None, 2).backward(dU.float().permute(0, 3, 1, 2))
    ref = xr.grad.permute(0, 2, 3, 1)
    # the strided scatter only visits the even positions: residual and mask apply there, the rest stays zero
    sel = torch.zeros(N, H, W, 1, dtype=torch.bool, device=DEV)
    sel[:, ::2, ::2] = True
    ref = torch.where(sel, (ref + res.float()) * (xin.float() > 0), torch.zeros_like(ref))
    assert float((dx.float() - ref).abs().max()) <= 1e-2 * float(ref.abs().max()) + 1e-3


@pytest.mark.parametrize('case', [('1x1_plain', 2, 19, 23, 128, 256, 1, 1, 1), ('3x3_d2', 2, 17, 19, 128, 128, 3, 2, 1),
                                  ('3x3_s2', 2, 21, 23, 64, 64, 3, 1, 2), ('1x1_s2', 2, 21, 23, 256, 128, 1, 1, 2)],
                         ids=lambda c: c[0])
def test_conv_wgrad_batchnorm_affine_side_outputs(ops, case):
    """wdot[co] = <W[.][co][.], G[.][co][.]> with G the unscaled weight gradient, dbeta[co] = sum_p dU[p][co]: what the
    gradient of a trainable BatchNorm affine behind the convolution is assembled from (cms_wgrad_desc)."""
    name, N, H, W, Cin, Cout, k, dil, stride = case
    g = torch.Generator(device=DEV).manual_seed(31)
    pad = dil * (k - 1) // 2
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    x = _mk((N, H, W, Cin), g)
    du = _mk((N, Ho, Wo, Cout), g)
    w = _mk((Cout, Cin, k, k), g, (1.0 / (Cin * k * k)) ** 0.5)
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    dw = torch.zeros(k * k, Cout, Cin, device=DEV)
    wdot = torch.zeros(Cout, device=DEV)
    dbeta = torch.zeros(Cout, device=DEV)
    ops.conv_wgrad(du, x, ops.conv_taps(k, k, dil, pad), dw, stride=stride, scale=scale, w_bf16=_pack(w), wdot=wdot,
                   dbeta=dbeta)
    xr = x.float().permute(0, 3, 1, 2)
    wr = w.float().clone().requires_grad_(True)
    F.conv2d(xr, wr, None, stride, pad, dil).backward(du.float().permute(0, 3, 1, 2))
    G = wr.grad                                                       # (Cout, Cin, k, k), unscaled
    ref_dot = (G * w.float()).sum(dim=(1, 2, 3))
    ref_beta = du.float().sum(dim=(0, 1, 2))
    assert float((wdot - ref_dot).abs().max()) <= 2e-3 * float(ref_dot.abs().max()) + 1e-3
    assert float((dbeta - ref_beta).abs().max()) <= 2e-3 * float(ref_beta.abs().max()) + 1e-3
    ref_dw = (G * scale.view(-1, 1, 1, 1)).permute(2, 3, 0, 1).reshape(k * k, Cout, Cin)
    assert float((dw - ref_dw).abs().max()) <= 2e-3 * float(ref_dw.abs().max()) + 1e-3


@pytest.mark.parametrize('hw', [(41, 41), (32, 48), (17, 18)])
def test_strided_3x3_dgrad_as_four_phases(ops, hw):
    """Data gradient of the stride-2 3x3 convolution (DeepLab v3+ backbone, layer2.0.conv2) as the four phases of a
    transposed convolution on the MFMA kernel (backbone_hip._dgrad_strided) vs autograd of the same convolution."""
    import types
    from cutmix_semisup_seg_amd.backbone_hip import DeepLabV3PlusBackboneExecutor
    H, W = hw
    N, Cin, Cout = 2, 128, 128
    g = torch.Generator(device=DEV).manual_seed(H)
    w = _mk((Cout, Cin, 3, 3), g, 0.05)
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    du = _mk((N, Ho, Wo, Cout), g)
    mask = _mk((N, H, W, Cin), g)
    wT = ops.conv_pack_transpose(_pack(w), scale=scale, flip=False)
    c = types.SimpleNamespace(stride=2, ksize=3, pad=1, dil=1, cin=Cin, wT=wT)
    ex = types.SimpleNamespace(_phase_w={})          # (the executor keeps the phases' sub-weights in persistent buffers)
    dx = DeepLabV3PlusBackboneExecutor._dgrad_strided(ex, du, c, mask, (H, W))
    xr = torch.zeros(N, Cin, H, W, device=DEV, requires_grad=True)
    y = F.conv2d(xr, w.float(), None, 2, 1) * scale.view(1, -1, 1, 1)
    y.backward(du.float().permute(0, 3, 1, 2))
    ref = xr.grad.permute(0, 2, 3, 1) * (mask.float() > 0)
    assert float((dx.float() - ref).abs().max()) <= 1.5e-2 * float(ref.abs().max()) + 1e-3


@pytest.mark.parametrize('case', [('l3_1x1_1024_256', 20, 41, 41, 1024, 256, 1, 1, 'fwd'),
                                  ('l3_3x3_d2', 20, 41, 41, 256, 256, 3, 2, 'fwd_res'),
                                  ('l3_1x1_256_1024', 20, 41, 41, 256, 1024, 1, 1, 'fwd_res'),
                                  ('dgrad_mask_res', 20, 41, 41, 256, 256, 3, 2, 'dgrad'),
                                  ('c3_l3_1x1', 8, 65, 129, 512, 256, 1, 1, 'fwd')], ids=lambda c: c[0])
def test_conv_balanced_launch_equals_plain_tiles(ops, case):
    """Grids of a few workgroups more than a multiple of the 256 CUs run their last pixel tiles as 32-channel slices
    (conv_igemm_mixed_kernel, the automatic choice of cms_conv_igemm at the BASELINE shapes: 526 / 1052 / 2104
    workgroups). Same MFMA sequence per output element, so the result must equal the plain 128 x 128 launch
    (tile = 128 asks for it explicitly) bit for bit -- and the fp32 reference within bf16 rounding."""
    name, N, H, W, Cin, Cout, k, dil, kind = case
    g = torch.Generator(device=DEV).manual_seed(11)
    pad = dil * (k - 1) // 2
    x = _mk((N, H, W, Cin), g)
    w = _mk((Cout, Cin, k, k), g, (2.0 / (Cin * k * k)) ** 0.5)
    taps = ops.conv_taps(k, k, dil, pad)
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    bias = torch.randn(Cout, generator=g, device=DEV) * 0.1
    res = _mk((N, H, W, Cout), g) if kind != 'fwd' else None
    if kind == 'dgrad':
        act = _mk((N, H, W, Cout), g)
        kw = dict(res=res, mask_src=act, mode=1)
    else:
        kw = dict(scale=scale, bias=bias, res=res, relu=True)
    y_auto = ops.conv_igemm(x, _pack(w), taps, **kw)
    y_plain = ops.conv_igemm(x, _pack(w), taps, tile=128, **kw)
    assert torch.equal(y_auto, y_plain)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, 1, pad, dil).permute(0, 2, 3, 1)
    if kind == 'dgrad':
        ref = (ref + res.float()) * (act.float() > 0)
    else:
        ref = ref * scale + bias
        if res is not None:
            ref = ref + res.float()
        ref = ref.relu()
    torch.testing.assert_close(y_auto.float(), ref, rtol=2 ** -7, atol=2e-2)


def test_grouped_weight_gradients_equal_the_per_layer_launches(ops):
    """ops.conv_wgrad_group: the weight gradients of many layers as one grid per kind (pointwise / with taps) -- the same sums
    as the per-layer launches (fp32 atomics: the order of the additions differs), including the launches it has to issue one by
    one (64-channel layers, a single launch of its kind), accumulating INTO the gradient buffers."""
    g = torch.Generator(device=DEV).manual_seed(21)
    N, H, W = 3, 23, 19
    specs = [(256, 128, 1, 1, 1), (128, 128, 3, 2, 1), (128, 512, 1, 1, 1), (512, 128, 1, 1, 1), (128, 128, 3, 1, 1),
             (64, 128, 1, 1, 1), (256, 256, 3, 4, 1), (128, 256, 1, 1, 2)]
    jobs, refs = [], []
    for cin, cout, k, dil, stride in specs:
        pad = dil * (k - 1) // 2
        ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        x = _mk((N, H, W, cin), g)
        du = _mk((N, ho, wo, cout), g)
        scale = torch.rand(cout, generator=g, device=DEV) + 0.5
        taps = ops.conv_taps(k, k, dil, pad)
        init = torch.randn(k * k, cout, cin, generator=g, device=DEV)
        ref = init.clone()
        ops.conv_wgrad(du, x, taps, ref, stride=stride, scale=scale)
        dw = init.clone()
        jobs.append((du, x, taps, dw, stride, scale))
        refs.append(ref)
    launched = ops.conv_wgrad_group(jobs)
    assert launched == 2                                   # one grid of pointwise launches, one of launches with taps / strides
    torch.cuda.synchronize()
    for (du, x, taps, dw, stride, scale), ref in zip(jobs, refs):
        tol = 2e-5 * float(ref.abs().max())
        assert float((dw - ref).abs().max()) <= tol, (tuple(dw.shape), float((dw - ref).abs().max()), tol)


# ------------------------------------------------------------------------------------------------- ReLU masks as bits
@pytest.mark.parametrize('shape', [(2, 19, 23, 64, 256, 1), (3, 41, 41, 256, 1024, 1), (2, 17, 21, 128, 96, 1), (1, 33, 35, 64, 64, 3),
                                   (20, 41, 41, 256, 1024, 1)],
                         ids=['256 out', 'expansion', '96 out (32-channel tiles)', '3x3 64 out', 'cfg 2 expansion (balanced launch)'])
def test_relu_mask_bits_written_by_the_forward_launch_and_read_by_the_data_gradient(ops, shape):
    """cms_conv_desc.mask_bits_out / mask_bits: the forward + ReLU launch also writes [y > 0] as bits (of the stored bf16
    value), the data gradient that needs that activation only for its sign takes the bits instead of re-reading it -- same
    result bit for bit, with and without the gradient of the residual branch."""
    N, H, W, Cin, Cout, k = shape
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
    w = (torch.randn(k * k, Cout, Cin, generator=g, device=DEV) * (1.0 / (Cin * k * k)) ** 0.5).bfloat16()
    taps = ops.conv_taps(k, k, 1, (k - 1) // 2)
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    bias = torch.randn(Cout, generator=g, device=DEV) * 0.2
    res = torch.randn(N, H, W, Cout, generator=g, device=DEV).bfloat16()
    for r in (None, res):
        y_ref = ops.conv_igemm(x, w, taps, scale=scale, bias=bias, res=r, relu=True, variant=99)
        bits = torch.full((N, H, W, Cout // 8), 0xAA, dtype=torch.uint8, device=DEV)
        y = ops.conv_igemm(x, w, taps, scale=scale, bias=bias, res=r, relu=True, mask_bits_out=bits)
        torch.cuda.synchronize()
        assert torch.equal(y, y_ref)
        want = (y.float() > 0).view(N, H, W, Cout // 8, 8).to(torch.int32)
        want = (want * (2 ** torch.arange(8, device=DEV, dtype=torch.int32))).sum(-1).to(torch.uint8)
        assert torch.equal(bits, want)
    # the data gradient of a convolution whose INPUT is y (Cout channels in, any width out)
    Cd = 256
    du = (torch.randn(N, H, W, Cd, generator=g, device=DEV) * 0.1).bfloat16()
    wT = (torch.randn(1, Cout, Cd, generator=g, device=DEV) * 0.05).bfloat16()          # [tap][ci = Cout of y][co = Cd]
    add = torch.randn(N, H, W, Cout, generator=g, device=DEV).bfloat16()
    one = ops.conv_taps(1, 1, 1, 0)
    for r in (None, add):
        a = ops.conv_igemm(du, wT, one, res=r, mode=1, mask_src=y, variant=99)
        b = ops.conv_igemm(du, wT, one, res=r, mode=1, mask_bits=bits)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        ops.conv_igemm(du, wT, one, mode=1, mask_src=y, mask_bits=bits)


def test_pack_transpose_plan_64_tiles_equal_the_32_tile_kernel_and_torch(monkeypatch):
    """`ops.PackTransposePlan`: wT[tap][ci][co] = bf16(w[tap][co][ci] * scale[co]) for many tensors in one launch. Round 6: tensors
    whose channel counts are multiples of 64 go through 64 x 64 tiles with 16-byte accesses (cms_conv_pack_transpose_batch64) -- bit
    for bit the result of the 32 x 32 kernel and of the same arithmetic in torch; a plan with an unaligned tensor keeps the old kernel."""
    from cutmix_semisup_seg_amd import ops
    g = torch.Generator(device=DEV).manual_seed(4)
    shapes = [(1, 256, 1024), (9, 256, 256), (1, 2048, 512), (1, 384, 2048), (3, 64, 128)]

    def make(shapes_):
        tr = []
        for i, (t, co, ci) in enumerate(shapes_):
            w = torch.randn(t, co, ci, generator=g, device=DEV).bfloat16()
            sc = (0.5 + torch.rand(co, generator=g, device=DEV)) if i % 2 == 0 else None
            tr.append((w, torch.zeros(t, ci, co, dtype=torch.bfloat16, device=DEV), sc))
        return tr

    def want(w, sc):
        v = w.float() * (sc.view(1, -1, 1) if sc is not None else 1.0)
        return v.bfloat16().permute(0, 2, 1).contiguous()
    tr = make(shapes)
    plan = ops.PackTransposePlan(tr)
    assert plan.tile64
    plan.run()
    out64 = [d.clone() for _, d, _ in tr]
    for (w, _, sc), o in zip(tr, out64):
        assert torch.equal(o, want(w, sc))
    monkeypatch.setenv('CMS_PACK64', '0')
    for _, d, _ in tr:
        d.zero_()
    plan32 = ops.PackTransposePlan(tr)
    assert not plan32.tile64
    plan32.run()
    assert all(torch.equal(d, o) for (_, d, _), o in zip(tr, out64))
    monkeypatch.delenv('CMS_PACK64')
    odd = make(shapes[:2] + [(1, 48, 144)])
    p3 = ops.PackTransposePlan(odd)
    assert not p3.tile64
    p3.run()
    for w, d, sc in odd:
        assert torch.equal(d, want(w, sc))
