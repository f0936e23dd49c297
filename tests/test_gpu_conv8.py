"""
GPU parity of the eight-phase 256 x 256 convolution (csrc/conv8.hip: cms_conv_igemm variants 90 = one whole tile per
workgroup, 91 = persistent launch with a stream-K round) through the C ABI:
  * against the same operator in FP64 ON THE HOST on identical bf16 inputs -- one bf16 rounding of the output plus fp32
    accumulation noise, every epilogue kind the backbone uses (plain / BN affine + ReLU / + residual; data gradient with
    ReLU mask, with mask AND gradient add, with the add alone), 1 x 1 / dilated 3 x 3 / odd K-tile counts / ragged pixel
    tiles / a strided output scatter;
  * against the default 128 x 128 kernel: whole tiles accumulate in the same K order, so variant 90 is BIT-IDENTICAL to it;
  * the stream-K round sums a tile's pieces in run order: repeated launches are bit-identical and within one bf16 rounding
    of the default kernel.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
BF16_HALF_ULP = 2.0 ** -8


@pytest.fixture(scope='module')
def ops():
    from cutmix_semisup_seg_amd import ops as _ops
    return _ops


def _mk(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen, device=DEV) * scale).bfloat16()


def _d(t):
    return t.detach().double().cpu()


def _pack(w):          # (Cout, Cin, kh, kw) bf16 -> (taps, Cout, Cin)
    co, ci, kh, kw = w.shape
    return w.permute(2, 3, 0, 1).reshape(kh * kw, co, ci).contiguous()


CASES = [
    # name, N, H, W, Cin, Cout, k, dil
    ('l3_conv1_1x1_1024_256', 2, 41, 41, 1024, 256, 1, 1),
    ('l3_conv2_3x3_d2', 2, 41, 41, 256, 256, 3, 2),
    ('l4_conv2_3x3_d4', 1, 29, 31, 512, 512, 3, 4),
    ('odd_k_tiles_3x3_c64', 2, 17, 23, 64, 256, 3, 2),
    ('one_ragged_tile', 1, 9, 13, 128, 256, 3, 1),
    ('tiles_5x3', 3, 20, 21, 192, 768, 3, 3),
]


def _close_to_fp64(y, ref):
    err = (_d(y) - ref).abs()
    tol = 1.02 * BF16_HALF_ULP * ref.abs() + 1e-5 * float(ref.abs().max())
    assert bool((err <= tol).all()), 'max err {} at scale {}'.format(float(err.max()), float(ref.abs().max()))
    assert float(err.mean()) <= 0.5 * BF16_HALF_ULP * float(ref.abs().mean()) + 1e-6


@pytest.mark.parametrize('variant', [90, 91])
@pytest.mark.parametrize('epi', ['plain', 'bn_relu', 'bn_res_relu'])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_conv8_forward_vs_fp64(ops, case, epi, variant):
    name, N, H, W, Cin, Cout, k, dil = case
    g = torch.Generator(device=DEV).manual_seed(len(name) * 7 + k)
    pad = dil * (k - 1) // 2
    x = _mk((N, H, W, Cin), g)
    w = _mk((Cout, Cin, k, k), g, (2.0 / (Cin * k * k)) ** 0.5)
    scale = (torch.rand(Cout, generator=g, device=DEV) + 0.5) if epi != 'plain' else None
    bias = torch.randn(Cout, generator=g, device=DEV) * 0.1 if epi != 'plain' else None
    res = _mk((N, H, W, Cout), g) if epi == 'bn_res_relu' else None
    relu = epi != 'plain'
    taps = ops.conv_taps(k, k, dil, pad)
    y = ops.conv_igemm(x, _pack(w), taps, scale=scale, bias=bias, res=res, relu=relu, variant=variant)
    ref = F.conv2d(_d(x).permute(0, 3, 1, 2), _d(w), None, 1, pad, dil)
    if scale is not None:
        ref = ref * _d(scale).view(1, -1, 1, 1) + _d(bias).view(1, -1, 1, 1)
    if res is not None:
        ref = ref + _d(res).permute(0, 3, 1, 2)
    if relu:
        ref = F.relu(ref)
    _close_to_fp64(y, ref.permute(0, 2, 3, 1))
    y0 = ops.conv_igemm(x, _pack(w), taps, scale=scale, bias=bias, res=res, relu=relu, variant=99)     # the 128 x 128 kernel
    if variant == 90:
        assert torch.equal(y, y0), 'whole tiles must equal the default kernel bit for bit'
    else:
        for _ in range(2):      # run-order sums: reproducible
            assert torch.equal(ops.conv_igemm(x, _pack(w), taps, scale=scale, bias=bias, res=res, relu=relu, variant=variant), y)


@pytest.mark.parametrize('variant', [90, 91])
@pytest.mark.parametrize('epi', ['mask', 'mask_add', 'add'])
@pytest.mark.parametrize('case', [CASES[0], CASES[1], CASES[3]], ids=[CASES[0][0], CASES[1][0], CASES[3][0]])
def test_conv8_data_gradient_vs_fp64(ops, case, epi, variant):
    """dX = (conv^T(dU * scale) + add) * [x > 0] through the same kernel on the packed-transposed weights."""
    name, N, H, W, Cin, Cout, k, dil = case          # here the GEMM runs Cout -> Cin: its output channels are Cin
    if Cin % 256 != 0:
        Cin = 256
    g = torch.Generator(device=DEV).manual_seed(31 + k)
    pad = dil * (k - 1) // 2
    x = torch.relu(_mk((N, H, W, Cin), g))
    w = _mk((Cout, Cin, k, k), g, (2.0 / (Cin * k * k)) ** 0.5)
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    dU = _mk((N, H, W, Cout), g)
    add = _mk((N, H, W, Cin), g) if epi != 'mask' else None
    mask = x if epi != 'add' else None
    wT = ops.conv_pack_transpose(_pack(w), scale=scale, flip=True)      # (taps, Cin, Cout)
    dx = ops.conv_igemm(dU, wT, ops.conv_taps(k, k, dil, pad), res=add, mode=1, mask_src=mask, variant=variant)
    xr = _d(x).permute(0, 3, 1, 2).requires_grad_(True)
    ws = (w.float() * scale.view(-1, 1, 1, 1)).bfloat16()      # the kernel's operand is bf16(w * scale)
    yy = F.conv2d(xr, _d(ws), None, 1, pad, dil)
    ref, = torch.autograd.grad(yy, xr, _d(dU).permute(0, 3, 1, 2))
    ref = ref.permute(0, 2, 3, 1)
    if add is not None:
        ref = ref + _d(add)
    if mask is not None:
        ref = ref * (_d(x) > 0)
    _close_to_fp64(dx, ref.detach())
    if variant == 90:
        assert torch.equal(dx, ops.conv_igemm(dU, wT, ops.conv_taps(k, k, dil, pad), res=add, mode=1, mask_src=mask, variant=99))


def test_conv8_strided_scatter_equals_default(ops):
    """A stride-2 data gradient phase: the GEMM grid is the dy grid, outputs are scattered to every second pixel."""
    g = torch.Generator(device=DEV).manual_seed(3)
    N, h, w_, Cin, Cout = 2, 21, 19, 128, 256
    du = _mk((N, h, w_, Cin), g)
    wt = _mk((4, Cout, Cin), g, 0.05)
    taps = [(0, 0), (0, 1), (1, 0), (1, 1)]
    outs = []
    for var in (99, 90, 91):
        out = torch.zeros(N, 2 * h, 2 * w_, Cout, dtype=torch.bfloat16, device=DEV)
        ops.conv_igemm(du, wt, taps, mode=1, out=out, out_hw=(h, w_), out_stride=2, out_full_hw=(2 * h, 2 * w_), variant=var)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert float((outs[2].float() - outs[0].float()).abs().max()) <= 2.0 ** -7 * float(outs[0].float().abs().max())


def test_conv8_two_streams_with_their_own_workspaces(ops):
    """Student || teacher: two streams run the stream-K variant at once; eager launches get one workspace per stream."""
    case = CASES[1]
    name, N, H, W, Cin, Cout, k, dil = case
    g = torch.Generator(device=DEV).manual_seed(17)
    taps = ops.conv_taps(k, k, dil, dil)
    xs = [_mk((4 * N, H, W, Cin), g) for _ in range(2)]
    ws = [_pack(_mk((Cout, Cin, k, k), g, 0.03)) for _ in range(2)]
    refs = [ops.conv_igemm(xs[i], ws[i], taps, relu=True, variant=99) for i in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for _ in range(4):
        outs = [torch.full_like(r, 7.0) for r in refs]
        torch.cuda.synchronize()
        for _ in range(3):
            for i in range(2):
                with torch.cuda.stream(streams[i]):
                    ops.conv_igemm(xs[i], ws[i], taps, relu=True, out=outs[i], variant=91)
        torch.cuda.synchronize()
        for i in range(2):
            d = (outs[i].float() - refs[i].float()).abs()
            assert bool((d <= 2.0 ** -7 * torch.maximum(outs[i].float().abs(), refs[i].float().abs()) + 1e-5).all())


# ------------------------------------------------------------------------------------------------- ReLU masks as bits (round 5)
@pytest.mark.parametrize('shape', [(2, 41, 41, 1024, 256, 1, 1), (2, 41, 41, 256, 256, 3, 2), (1, 9, 13, 128, 512, 3, 1),
                                   (3, 20, 21, 192, 768, 3, 3)],
                         ids=['l3 conv1', 'l3 conv2', 'ragged tile, 2 channel tiles', 'tiles 5x3'])
def test_conv8_relu_mask_bits_written_forward_and_read_by_the_data_gradient(ops, shape):
    """cms_conv_desc.mask_bits_out / mask_bits on the eight-phase kernel: same bit layout as the 128 x 128 kernel's (one dword
    per pixel row and 32 channels, bit = channel), outputs bit-identical to the launches that take / re-read the activation."""
    N, H, W, Cin, Cout, k, dil = shape
    g = torch.Generator(device=DEV).manual_seed(31 + Cin)
    pad = dil * (k - 1) // 2
    x = _mk((N, H, W, Cin), g)
    w = _pack(_mk((Cout, Cin, k, k), g, (2.0 / (Cin * k * k)) ** 0.5))
    taps = ops.conv_taps(k, k, dil, pad)
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    bias = torch.randn(Cout, generator=g, device=DEV) * 0.2
    res = _mk((N, H, W, Cout), g)
    for r in (None, res):
        y_ref = ops.conv_igemm(x, w, taps, scale=scale, bias=bias, res=r, relu=True, variant=99)
        bits = torch.full((N, H, W, Cout // 8), 0xAA, dtype=torch.uint8, device=DEV)
        y = ops.conv_igemm(x, w, taps, scale=scale, bias=bias, res=r, relu=True, mask_bits_out=bits, variant=90)
        torch.cuda.synchronize()
        assert torch.equal(y, y_ref)
        want = (y.float() > 0).view(N, H, W, Cout // 8, 8).to(torch.int32)
        want = (want * (2 ** torch.arange(8, device=DEV, dtype=torch.int32))).sum(-1).to(torch.uint8)
        assert torch.equal(bits, want)
        bits128 = torch.zeros_like(bits)
        ops.conv_igemm(x, w, taps, scale=scale, bias=bias, res=r, relu=True, mask_bits_out=bits128, variant=0, tile=128)
        torch.cuda.synchronize()
        assert torch.equal(bits, bits128)                 # the two kernels of cms_conv_igemm write the same bits
    # the data gradient of a convolution whose INPUT is y: K = Cd channels of du -> Cout channels of dx, masked by [y > 0]
    Cd = 1024
    du = _mk((N, H, W, Cd), g, 0.1)
    wT = _mk((1, Cout, Cd), g, 0.05)
    add = _mk((N, H, W, Cout), g)
    one = ops.conv_taps(1, 1, 1, 0)
    for r in (None, add):
        a = ops.conv_igemm(du, wT, one, res=r, mode=1, mask_src=y, variant=99)
        b = ops.conv_igemm(du, wT, one, res=r, mode=1, mask_bits=bits, variant=90)
        c = ops.conv_igemm(du, wT, one, res=r, mode=1, mask_bits=bits)                    # the library's own choice of kernel
        torch.cuda.synchronize()
        assert torch.equal(a, b) and torch.equal(a, c)
    # (round 5) mask_gates_res: the bits gate the RESIDUAL only -- acc + (bit ? res : 0), the masked gradient of an identity
    # shortcut added in the epilogue; == the plain residual launch on a pre-masked residual, on every kernel
    gated = torch.where(y > 0, add, torch.zeros_like(add))
    for kw_cd, wT_ in ((Cd, wT), (64, _mk((1, Cout, 64), g, 0.05))):     # K = 16 tiles: eight-phase kernel; K = 1 tile: 128 x 128 family
        du_ = du if kw_cd == Cd else _mk((N, H, W, 64), g, 0.1)
        ref = ops.conv_igemm(du_, wT_, one, res=gated, mode=1, variant=99)
        got = ops.conv_igemm(du_, wT_, one, res=add, mode=1, mask_bits=bits, mask_gates_res=True)
        got99 = ops.conv_igemm(du_, wT_, one, res=add, mode=1, mask_bits=bits, mask_gates_res=True, variant=99)
        torch.cuda.synchronize()
        assert torch.equal(ref, got) and torch.equal(ref, got99)
    got90 = ops.conv_igemm(du, wT, one, res=add, mode=1, mask_bits=bits, mask_gates_res=True, variant=90)
    assert torch.equal(ops.conv_igemm(du, wT, one, res=gated, mode=1, variant=99), got90)
    with pytest.raises(ValueError):
        ops.conv_igemm(du, wT, one, mode=1, mask_bits=bits, mask_gates_res=True)          # no residual to gate
    with pytest.raises(ValueError):
        ops.conv_igemm(du, wT, one, mode=1, mask_bits=bits, variant=91)                   # stream-K launch: no bits
