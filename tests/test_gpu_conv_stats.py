"""
GPU: BatchNorm statistics out of the convolution epilogue (round 5; cms_conv_desc.stats_out, csrc/tile_stats.hpp,
cms_bn_finalize_tiles) -- the batch-statistics units of architectures/deeplab2.py:72-84 (reference: torchvision-style
Bottleneck with nn.BatchNorm2d in training mode, train_seg_semisup_mask_mt.py:587 default) without the pass over u.

  * the launch with stats_out stores EXACTLY the output of the launch without it (bit for bit), on every kernel that takes the
    default routes (eight-phase 256 x 256, balanced 128 x 128 + 32-channel slices, 128 / 64 / 32-channel tiles);
  * the tile sums equal fp64 sums of the stored bf16 values per tile and slot (sample groups: a tile straddling a group boundary
    keeps the two groups apart), within fp32 accumulation error, and repeat bit for bit;
  * cms_bn_finalize_tiles gives the mean / rstd / scale / shift / running statistics / batch counter of cms_bn_stats on the same u.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

# (name, N, H, W, Cin, Cout, kernel, dilation, expected tile rows)
CASES = [
    ('conv8_1x1', 4, 41, 41, 1024, 256, 1, 1, 256),        # K tiles = 16: eight-phase kernel; M = 6724 = 26.3 tiles
    ('conv8_3x3', 2, 33, 29, 256, 512, 3, 2, 256),         # dilated 3 x 3, K tiles = 36
    ('mixed_128', 20, 41, 41, 64, 256, 1, 1, 128),         # 263 x 2 = 526 tiles = 2 * 256 + 14: the balanced launch
    ('tile128', 2, 23, 19, 128, 128, 3, 1, 128),
    ('tile64', 2, 47, 31, 64, 64, 3, 1, 128),
    ('tile32', 2, 21, 17, 64, 32, 1, 1, 128),
    ('tile64_2048_tiles', 4, 256, 256, 64, 64, 1, 1, 128),  # more tiles than the finalising launch holds in registers (1280)
]


def _run(name, N, H, W, Cin, Cout, k, dil, rows, G):
    from cutmix_semisup_seg_amd import ops
    if (N * H * W) % G != 0 or (N * H * W) // G < rows:
        pytest.skip('{} sample groups do not fit this case'.format(G))
    g = torch.Generator().manual_seed(len(name) * 131 + G)
    x = (torch.randn(N, H, W, Cin, generator=g) * 0.8 + 0.1).to(torch.bfloat16).to(DEV)
    w = (torch.randn(k * k, Cout, Cin, generator=g) * (1.5 / np.sqrt(Cin * k * k))).to(torch.bfloat16).to(DEV)
    taps = ops.conv_taps(k, k, dil, dil * (k // 2))
    plain = ops.conv_igemm(x, w, taps)
    st = {'groups': G}
    u = ops.conv_igemm(x, w, taps, stats=st)
    torch.cuda.synchronize()
    return ops, x, w, taps, plain, u, st


@pytest.mark.parametrize('G', [1, 2, 4])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_epilogue_tile_sums_match_fp64_sums_of_the_stored_output(case, G):
    name, N, H, W, Cin, Cout, k, dil, rows = case
    ops, x, w, taps, plain, u, st = _run(*case, G)
    assert st['tile_rows'] == rows, 'launch took another kernel than this case is meant for'
    assert torch.equal(plain.view(torch.int16), u.view(torch.int16))        # the statistics do not change what is stored
    M = N * H * W
    RG = M // G
    T = rows
    nt = (M + T - 1) // T
    ts = st['tile_sums'].view(nt, 2, 2, Cout).double().cpu()
    flat = u.view(M, Cout).double().cpu()
    for t in range(nt):
        lo, hi = t * T, min((t + 1) * T, M)
        b = (lo // RG + 1) * RG                       # first row of the next sample group
        parts = [(0, lo, min(hi, b))] + ([(1, b, hi)] if b < hi else [])
        for slot, a, e in parts:
            blk = flat[a:e]
            want_s, want_q = blk.sum(0), (blk * blk).sum(0)
            tol_s = 1e-5 * blk.abs().sum(0) + 1e-6
            assert ((ts[t, slot, 0] - want_s).abs() <= tol_s).all(), (name, t, slot)
            assert ((ts[t, slot, 1] - want_q).abs() <= 1e-5 * want_q + 1e-6).all(), (name, t, slot)
    # fixed order: bit-reproducible
    st2 = {'groups': G}
    ops.conv_igemm(x, w, taps, stats=st2)
    torch.cuda.synchronize()
    written = torch.zeros(nt, 2, dtype=torch.bool)
    for t in range(nt):
        lo, hi = t * T, min((t + 1) * T, M)
        written[t, 0] = True
        written[t, 1] = (lo // RG + 1) * RG < hi
    a = st['tile_sums'].view(nt, 2, 2 * Cout).cpu()[written]
    b2 = st2['tile_sums'].view(nt, 2, 2 * Cout).cpu()[written]
    assert torch.equal(a.view(torch.int32), b2.view(torch.int32))


@pytest.mark.parametrize('G', [1, 2, 4])      # 4: three boundaries, three straddling tiles (their slot-1 sums come from lanes 0 .. 2)
@pytest.mark.parametrize('case', [CASES[0], CASES[2], CASES[4], CASES[6]], ids=[CASES[0][0], CASES[2][0], CASES[4][0], CASES[6][0]])
def test_finalize_tiles_equals_the_statistics_pass_over_the_output(case, G):
    name, N, H, W, Cin, Cout, k, dil, rows = case
    ops, x, w, taps, plain, u, st = _run(*case, G)
    M = N * H * W
    g = torch.Generator().manual_seed(7)
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).to(DEV), (torch.randn(Cout, generator=g) * 0.2).to(DEV)

    def outputs():
        return dict(mean=torch.empty(G * Cout, device=DEV), rstd=torch.empty(G * Cout, device=DEV),
                    scale=torch.empty(G * Cout, device=DEV), shift=torch.empty(G * Cout, device=DEV),
                    running_mean=torch.full((Cout,), 0.25, device=DEV), running_var=torch.full((Cout,), 1.5, device=DEV),
                    counter=torch.full((), 3, dtype=torch.int64, device=DEV))
    a, b = outputs(), outputs()
    ops.bn_op('stats', c=Cout, dtype=torch.bfloat16, n_pixels=M, groups=G, eps=1e-5, momentum=0.1, x=u,
              ws=ops.bn_workspace(M, Cout, DEV, G), gamma=gamma, beta=beta, **a)
    ops.bn_op('finalize_tiles', c=Cout, dtype=torch.bfloat16, n_pixels=M, groups=G, eps=1e-5, momentum=0.1,
              tile_rows=st['tile_rows'], ws=st['tile_sums'], gamma=gamma, beta=beta, **b)
    torch.cuda.synchronize()
    assert int(a['counter']) == int(b['counter']) == 3 + G
    for key in ('mean', 'rstd', 'scale', 'shift', 'running_mean', 'running_var'):
        torch.testing.assert_close(b[key], a[key], rtol=2e-5, atol=2e-6, msg=lambda m, k_=key: '{}: {}'.format(k_, m))
    # and against fp64 on the host, group by group
    flat = u.view(G, M // G, Cout).double().cpu()
    mean = flat.mean(1)
    var = flat.var(1, unbiased=False)
    torch.testing.assert_close(b['mean'].view(G, Cout).double().cpu(), mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(b['rstd'].view(G, Cout).double().cpu(), 1.0 / torch.sqrt(var + 1e-5), rtol=1e-5, atol=1e-6)


def test_launches_that_cannot_write_tile_sums_say_so():
    """Sample groups shorter than a tile, fp32 outputs and data gradients report tile_rows 0 (the caller keeps the pass over u);
    a descriptor that carries stats_out anyway is refused, not silently ignored."""
    from cutmix_semisup_seg_amd import ops, _lib
    x = torch.randn(2, 9, 9, 64, device=DEV).to(torch.bfloat16)           # M = 162, two groups of 81 rows < 128
    w = torch.randn(1, 64, 64, device=DEV).to(torch.bfloat16)
    st = {'groups': 2}
    ops.conv_igemm(x, w, [(0, 0)], stats=st)
    assert st['tile_rows'] == 0 and st['tile_sums'] is None
    st = {'groups': 1}
    ops.conv_igemm(x, w, [(0, 0)], stats=st)
    assert st['tile_rows'] == 128
    st = {'groups': 1}
    ops.conv_igemm(x, w, [(0, 0)], mode=1, stats=st)
    assert st['tile_rows'] == 0


# (name, N, H, W, K channels of du, channels of the gradient written, expected tile rows)
BWD_CASES = [
    ('conv8', 4, 41, 41, 1024, 256, 256),
    ('mixed_128', 20, 41, 41, 64, 256, 128),
    ('tile64', 2, 47, 31, 128, 64, 128),
]


@pytest.mark.parametrize('G', [1, 2, 4])
@pytest.mark.parametrize('relu,gated', [(True, False), (True, True), (False, False)])
@pytest.mark.parametrize('case', BWD_CASES, ids=[c[0] for c in BWD_CASES])
def test_backward_statistics_from_the_data_gradient_epilogue(case, G, relu, gated):
    """A data-gradient launch that writes the gradient dy of a batch-statistics unit's output also leaves (sum d, sum d xhat) per
    tile (cms_conv_desc.bstats_*): the stored dy is unchanged, cms_bn_bwd_sums_tiles gives the sums of the reduction kernel
    (cms_bn_reduce_ws mode 1 over u, dy and the mask bits) within fp32 accumulation error, bit-reproducibly."""
    from cutmix_semisup_seg_amd import ops
    name, N, H, W, Cd, C, rows = case
    g = torch.Generator().manual_seed(len(name) + 7 * G)
    M = N * H * W
    if M % G != 0 or M // G < rows:
        pytest.skip('{} sample groups do not fit this case'.format(G))
    du = (torch.randn(N, H, W, Cd, generator=g) * 0.3).to(torch.bfloat16).to(DEV)
    wT = (torch.randn(1, C, Cd, generator=g) * (1.0 / np.sqrt(Cd))).to(torch.bfloat16).to(DEV)
    u = (torch.randn(N, H, W, C, generator=g) * 1.1 + 0.4).to(torch.bfloat16).to(DEV)
    res = (torch.randn(N, H, W, C, generator=g) * 0.5).to(torch.bfloat16).to(DEV) if gated else None
    gate_bits = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.uint8).to(DEV) if gated else None
    bits = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.uint8).to(DEV) if relu else None
    # the unit's statistics (any consistent mean / rstd will do: take the real ones)
    flat = u.view(G, M // G, C).float()
    mean = flat.mean(1).reshape(-1).contiguous()
    rstd = (1.0 / torch.sqrt(flat.var(1, unbiased=False) + 1e-5)).reshape(-1).contiguous()
    one = [(0, 0)]
    plain = ops.conv_igemm(du, wT, one, res=res, mode=1, mask_bits=gate_bits, mask_gates_res=gated)
    st = {'groups': G, 'u': u, 'mean': mean, 'rstd': rstd, 'bits': bits}
    dy = ops.conv_igemm(du, wT, one, res=res, mode=1, mask_bits=gate_bits, mask_gates_res=gated, stats=st)
    torch.cuda.synchronize()
    assert st['tile_rows'] == rows
    assert torch.equal(plain.view(torch.int16), dy.view(torch.int16))
    got = torch.empty(G * 2 * C, dtype=torch.float64, device=DEV)
    ops.bn_op('sums_tiles', c=C, dtype=torch.bfloat16, n_pixels=M, groups=G, tile_rows=rows, ws=st['tile_sums'], sums=got)
    want = torch.empty(G * 2 * C, dtype=torch.float64, device=DEV)
    ws = ops.bn_workspace(M, C, DEV, G)
    if relu:
        ops.bn_op('reduce_bwd', c=C, dtype=torch.bfloat16, n_pixels=M, groups=G, x=u, dy=dy, mean=mean, rstd=rstd, sums=want, ws=ws,
                  mask_bits=bits)
    else:
        ops.bn_op('reduce_bwd', c=C, dtype=torch.bfloat16, n_pixels=M, groups=G, x=u, dy=dy, y=None, mean=mean, rstd=rstd, sums=want,
                  ws=ws)
    torch.cuda.synchronize()
    # scale of the error: fp32 accumulation over <= 256 rows per tile of terms |d| and |d xhat|
    d = dy.view(G, M // G, C).double()
    if relu:
        on = ((bits.view(M, C // 8, 1).to(torch.int32) >> torch.arange(8, device=DEV, dtype=torch.int32)) & 1).view(G, M // G, C)
        d = d * on
    xh = (u.view(G, M // G, C).double() - mean.view(G, 1, C).double()) * rstd.view(G, 1, C).double()
    l1 = torch.stack([d.abs().sum(1), (d * xh).abs().sum(1)], 1).reshape(-1)
    exact = torch.stack([d.sum(1), (d * xh).sum(1)], 1).reshape(-1)
    assert ((got - exact).abs() <= 2e-6 * l1 + 1e-6).all()
    assert ((want - exact).abs() <= 2e-6 * l1 + 1e-6).all()
    st2 = {'groups': G, 'u': u, 'mean': mean, 'rstd': rstd, 'bits': bits}
    ops.conv_igemm(du, wT, one, res=res, mode=1, mask_bits=gate_bits, mask_gates_res=gated, stats=st2)
    again = torch.empty_like(got)
    ops.bn_op('sums_tiles', c=C, dtype=torch.bfloat16, n_pixels=M, groups=G, tile_rows=rows, ws=st2['tile_sums'], sums=again)
    torch.cuda.synchronize()
    assert torch.equal(got, again)
