"""
The eight-phase 256 x 256 weight gradient (csrc/wgrad8.hip; autograd of architectures/deeplab2.py:89-109 with respect to the
convolution weights) against an fp32 weight gradient of the same bf16 operands (ATen, on the device) and against the
128 x 128 kernel it replaces for the wide layers: taps, dilation, strides, a partial last K tile, pixel slices of odd
length, one / several slices, the deterministic slab route, and the dispatch rule (what falls back to the 128 x 128 kernel).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

CASES = [
    # name, N, H, W, Cin, Cout, k, dilation, stride
    ('1x1, partial last K tile', 3, 19, 21, 256, 256, 1, 1, 1),
    ('3x3 dilation 2', 2, 23, 37, 256, 256, 3, 2, 1),
    ('3x3 stride 2', 4, 33, 47, 256, 512, 3, 1, 2),
    ('1x1 stride 2 (shortcut)', 3, 41, 41, 512, 256, 1, 1, 2),
    ('3x3 dilation 4, two ci tiles', 1, 41, 41, 512, 256, 3, 4, 1),
    ('narrow map (Wo = 9)', 4, 40, 9, 256, 256, 3, 1, 1),
    ('layer3 1x1 of configs[1] at batch 4', 4, 41, 41, 1024, 256, 1, 1, 1),
]


@pytest.fixture(scope='module')
def ops():
    from cutmix_semisup_seg_amd import ops as o
    return o


@pytest.fixture()
def kernel_switch():
    from cutmix_semisup_seg_amd._lib import lib
    yield lib.cms_conv_set_wgrad8
    lib.cms_conv_set_wgrad8(-1)


def _make(ops, case, seed=0):
    name, N, H, W, Cin, Cout, k, dil, stride = case
    g = torch.Generator(device=DEV).manual_seed(seed)
    pad = dil * (k - 1) // 2
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
    du = (torch.randn(N, Ho, Wo, Cout, generator=g, device=DEV) * 0.05).bfloat16()
    scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
    return du, x, ops.conv_taps(k, k, dil, pad), scale, stride, (k, dil, pad)


def _reference(t):
    du, x, taps, scale, stride, (k, dil, pad) = t
    w = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (du.shape[3], x.shape[3], k, k), du.float().permute(0, 3, 1, 2),
                                    stride=stride, padding=pad, dilation=dil)
    return (w * scale.view(-1, 1, 1, 1)).permute(2, 3, 0, 1).reshape(k * k, du.shape[3], x.shape[3]).contiguous()


def _run(ops, t, ksplit=0, dw=None, wg_target=56):
    du, x, taps, scale, stride, _ = t
    if dw is None:
        dw = torch.zeros(len(taps), du.shape[3], x.shape[3], device=DEV)
    return ops.conv_wgrad(du, x, taps, dw, stride=stride, scale=scale, ksplit=ksplit, wg_target=wg_target)


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_eight_phase_weight_gradient_vs_fp32_and_the_128_tile_kernel(ops, kernel_switch, case):
    t = _make(ops, case)
    du, x, taps, scale, stride, _ = t
    ref = _reference(t)
    den = ref.abs().max().item()
    kernel_switch(0)
    assert ops.conv_wgrad(du, x, taps, torch.zeros(ref.shape, device=DEV), stride=stride, scale=scale, query_kernel=True, wg_target=56) == 0
    old = _run(ops, t)
    kernel_switch(1)
    assert ops.conv_wgrad(du, x, taps, torch.zeros(ref.shape, device=DEV), stride=stride, scale=scale, query_kernel=True, wg_target=56) == 8
    for ks in (0, 1, 3, 5):                     # automatic split, one slice (plain accumulation), odd slice lengths
        for rep in range(2):
            new = _run(ops, t, ksplit=ks)
            torch.cuda.synchronize()
            # fp32 accumulation of bf16 products in another order: 1e-5 of the tensor's range (measured 3e-7 ... 3e-6)
            assert (new - ref).abs().max().item() <= 1e-5 * den, (case[0], ks, rep)
            assert (new - old).abs().max().item() <= 1e-5 * den
    # accumulates INTO dw (the gradient arena adds the passes of an iteration up)
    acc = _run(ops, t, dw=ref.clone().contiguous())
    assert (acc - 2 * ref).abs().max().item() <= 2e-5 * den


def test_one_slice_is_bit_reproducible_and_slabs_make_several_slices_so(ops, kernel_switch):
    from cutmix_semisup_seg_amd import ops as o
    t = _make(ops, CASES[-1])
    kernel_switch(1)
    a, b = _run(ops, t, ksplit=1), _run(ops, t, ksplit=1)
    assert torch.equal(a, b)
    saved = o._WGRAD_DETERMINISTIC
    o._WGRAD_DETERMINISTIC = True
    try:
        outs = [_run(ops, t, ksplit=4) for _ in range(4)]
    finally:
        o._WGRAD_DETERMINISTIC = saved
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], v) for v in outs[1:])
    ref = _reference(t)
    assert (outs[0] - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_dispatch_rule(ops, kernel_switch):
    kernel_switch(1)
    q = lambda du, x, taps, **kw: ops.conv_wgrad(du, x, taps, torch.zeros(len(taps), du.shape[3], x.shape[3], device=DEV),
                                                 query_kernel=True, **kw)
    bf = lambda *s: torch.zeros(*s, device=DEV).bfloat16()
    one = ops.conv_taps(1, 1, 1, 0)
    assert q(bf(2, 33, 33, 256), bf(2, 33, 33, 512), one, wg_target=56) == 8  # beside other work: always
    assert q(bf(2, 33, 33, 256), bf(2, 33, 33, 512), one) == 0             # alone, 2 tiles: the atomics of a machine-wide split lose
    assert q(bf(2, 33, 33, 512), bf(2, 33, 33, 1024), one) == 8            # alone, 8 tiles
    assert q(bf(2, 33, 33, 128), bf(2, 33, 33, 512), one, wg_target=56) == 0             # Cout % 256
    assert q(bf(2, 33, 33, 256), bf(2, 33, 33, 128), one, wg_target=56) == 0             # Cin % 256
    assert q(bf(1, 20, 20, 256), bf(1, 20, 20, 256), one, wg_target=56) == 0             # fewer than 16 K tiles of pixels
    assert q(bf(64, 4, 4, 256), bf(64, 4, 4, 256), one, wg_target=56) == 0               # maps the branch-free cursor cannot walk
    wdot, dbeta = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    assert q(bf(2, 33, 33, 256), bf(2, 33, 33, 512), one, w_bf16=bf(1, 256, 512), wdot=wdot, dbeta=dbeta, wg_target=56) == 0   # BN-affine side outputs
    kernel_switch(0)
    assert q(bf(2, 33, 33, 256), bf(2, 33, 33, 512), one, wg_target=56) == 0
