"""
GPU: batch-statistics BatchNorm (+ residual, + ReLU) on csrc/bn.hip against nn.BatchNorm2d semantics in fp64 on the host
(forward, running statistics, every gradient), the library engine's training-mode layers routed through it, and the
synchronised variant (two ranks over gloo on one GPU: statistics all-reduced between the passes -- SURVEY.md 8(e)).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _reference(x, gamma, beta, res, relu, rm, rv, momentum, eps, dy):
    """fp64 host reference: x (N,H,W,C) -> y, new running stats, grads (dx, dgamma, dbeta, dres)."""
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rd = None if res is None else res.double().permute(0, 3, 1, 2).requires_grad_(True)
    rm2, rv2 = rm.double().clone(), rv.double().clone()
    y = F.batch_norm(xd, rm2, rv2, gd, bd, True, momentum, eps)
    if rd is not None:
        y = y + rd
    if relu:
        y = F.relu(y)
    y.backward(dy.double().permute(0, 3, 1, 2))
    nhwc = lambda t: t.permute(0, 2, 3, 1)
    return (nhwc(y.detach()), rm2, rv2, nhwc(xd.grad), gd.grad, bd.grad, None if rd is None else nhwc(rd.grad))


@pytest.mark.parametrize('C', [48, 64, 256, 2048, 2208])      # 2208 = DenseNet-161's norm5: two channel slices
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('relu,with_res', [(True, True), (False, False), (True, False)])
def test_batch_norm_act_vs_fp64(C, dtype, relu, with_res):
    from cutmix_semisup_seg_amd import ops
    g = torch.Generator().manual_seed(C + int(relu))
    N, H, W = 3, 13, 11
    x = (torch.randn(N, H, W, C, generator=g) * 1.7 + 0.3).to(dtype).float()
    res = torch.randn(N, H, W, C, generator=g).to(dtype).float() if with_res else None
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    dy = torch.randn(N, H, W, C, generator=g).to(dtype).float()
    want = _reference(x, gamma, beta, res, relu, rm, rv, 0.1, 1e-5, dy)
    cu = lambda t: None if t is None else t.to(DEV)
    xg = cu(x.to(dtype)).requires_grad_(True)
    gg, bg = cu(gamma).requires_grad_(True), cu(beta).requires_grad_(True)
    rg = None if res is None else cu(res.to(dtype)).requires_grad_(True)
    rmg, rvg = cu(rm.clone()), cu(rv.clone())
    y = ops.batch_norm_act(xg, gg, bg, rmg, rvg, 0.1, 1e-5, relu=relu, res=rg)
    y.backward(cu(dy.to(dtype)))
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    torch.testing.assert_close(y.detach().float().cpu(), want[0].float(), rtol=tol, atol=tol)
    torch.testing.assert_close(rmg.cpu(), want[1].float(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rvg.cpu(), want[2].float(), rtol=1e-5, atol=1e-6)
    if dtype == torch.bfloat16 and relu:
        return          # (a bf16-rounded output can flip a ReLU mask bit relative to the fp64 reference)
    gtol = 5e-4 if dtype == torch.float32 else 5e-2
    torch.testing.assert_close(xg.grad.float().cpu(), want[3].float(), rtol=gtol, atol=gtol * float(want[3].abs().max()))
    torch.testing.assert_close(gg.grad.cpu(), want[4].float(), rtol=gtol, atol=gtol * float(want[4].abs().max()))
    torch.testing.assert_close(bg.grad.cpu(), want[5].float(), rtol=gtol, atol=gtol * float(want[5].abs().max()))
    if with_res:
        torch.testing.assert_close(rg.grad.float().cpu(), want[6].float(), rtol=gtol, atol=gtol)


def test_layer_engine_training_mode_layers_use_the_hip_batchnorm():
    """DeepLab v2 WITHOUT --freeze_bn (BatchNorm on batch statistics, affine parameters frozen, deeplab2.py:72-84):
    forward, running statistics and gradients vs the oracle."""
    from architectures import deeplab2
    from oracle import deeplab2 as odl
    C, layers = 5, [1, 1, 1, 1]
    st = odl.closed_form_state(C, layers)
    net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    net.load_state_dict(st)
    net = net.to(DEV)
    net.compute_dtype = torch.float32
    net.train()                                            # no freeze_batchnorm()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(4, 3, 65, 65, generator=g)
    lo = net.forward_lowres(x.to(DEV))
    new_stats = {}
    want = odl.forward_lowres(x, st, layers, frozen=False, new_stats=new_stats)
    torch.testing.assert_close(lo.detach().cpu(), want, rtol=2e-3, atol=2e-4)
    sd = net.state_dict()
    for k in ('bn1.running_mean', 'layer2.0.bn2.running_var', 'layer4.0.downsample.1.running_mean'):
        torch.testing.assert_close(sd[k].cpu(), new_stats[k], rtol=1e-3, atol=1e-5)
    assert int(sd['layer3.0.bn1.num_batches_tracked']) == 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sync_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cutmix_semisup_seg_amd import ops
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)
        g = torch.Generator().manual_seed(5)
        C = 64
        x = torch.randn(4, 9, 7, C, generator=g)              # the WHOLE batch, identical on both ranks
        dy = torch.randn(4, 9, 7, C, generator=g)
        gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
        sl = slice(2 * rank, 2 * rank + 2)                    # this rank's shard
        xs = x[sl].contiguous().to(dev).requires_grad_(True)
        gg, bg = gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y = ops.batch_norm_act(xs, gg, bg, rm, rv, 0.1, 1e-5, relu=True)
        y.backward(dy[sl].contiguous().to(dev))
        q.put((rank, y.detach().cpu().numpy(), xs.grad.cpu().numpy(), gg.grad.cpu().numpy(), rm.cpu().numpy(),
               rv.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_sync_batchnorm_two_ranks_equals_single_process_on_the_whole_batch():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(5)
    C = 64
    x = torch.randn(4, 9, 7, C, generator=g)
    dy = torch.randn(4, 9, 7, C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    want = _reference(x, gamma, beta, None, True, torch.zeros(C), torch.ones(C), 0.1, 1e-5, dy)
    y = np.concatenate([res[0][1], res[1][1]], 0)
    dx = np.concatenate([res[0][2], res[1][2]], 0)
    np.testing.assert_allclose(y, want[0].float().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dx, want[3].float().numpy(), rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(res[0][3] + res[1][3], want[4].float().numpy(), rtol=5e-4, atol=5e-5)   # local dgamma's add up
    for r in res:                                                       # every rank holds the GLOBAL running statistics
        np.testing.assert_allclose(r[4], want[1].float().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r[5], want[2].float().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('shape', [(2, 41, 41, 1024), (3, 81, 81, 64), (1, 33, 65, 2208), (10, 41, 41, 256)],
                         ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_atomics_free_reductions_match_fp64_sums_and_repeat_bit_for_bit(shape, dtype):
    """cms_bn_reduce_ws (mode 0 and 1) and cms_bn_stats on multi-split geometries of the networks: sums against fp64 on the
    host, the fused finalisation against cms_bn_finalize on the same sums, the same workspace used over and over (the last
    block of a tile resets its counter), and two launches agreeing bit for bit (fixed summation order; the atomics kernel
    cms_bn_reduce is held to the same sums at its own rounding)."""
    from cutmix_semisup_seg_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    C = shape[-1]
    x = (torch.randn(*shape, generator=g) * 1.3 + 0.4).to(dtype)
    dy = torch.randn(*shape, generator=g).to(dtype)
    y = torch.relu(torch.randn(*shape, generator=g)).to(dtype)
    npix = x.numel() // C
    xd, dyd = x.double().reshape(npix, C), dy.double().reshape(npix, C)
    want0 = torch.cat([xd.sum(0), (xd * xd).sum(0)])
    mean = (want0[:C] / npix).float()
    rstd = (1.0 / torch.sqrt(want0[C:] / npix - (want0[:C] / npix) ** 2 + 1e-5)).float()
    dm = dyd * (y.double().reshape(npix, C) > 0)
    want1 = torch.cat([dm.sum(0), (dm * ((xd - mean.double()) * rstd.double())).sum(0)])
    xg, dyg, yg, mg, rg = (t.to(DEV) for t in (x, dy, y, mean, rstd))
    ws = ops.bn_workspace(npix, C, DEV)
    tol = dict(rtol=2e-5, atol=2e-3 * (npix ** 0.5) * 1e-2)

    def reduce(mode):
        out = torch.full((2 * C,), float('nan'), dtype=torch.float64, device=DEV)     # overwritten, not accumulated
        if mode == 0:
            ops.bn_op('reduce', c=C, dtype=dtype, n_pixels=npix, x=xg, sums=out, ws=ws)
        else:
            ops.bn_op('reduce_bwd', c=C, dtype=dtype, n_pixels=npix, x=xg, dy=dyg, y=yg, mean=mg, rstd=rg, sums=out, ws=ws)
        return out.cpu()

    a0, a1, b0, b1 = reduce(0), reduce(1), reduce(0), reduce(1)
    assert torch.equal(a0, b0) and torch.equal(a1, b1)
    torch.testing.assert_close(a0, want0, **tol)
    torch.testing.assert_close(a1, want1, **tol)
    old = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
    ops.bn_op('reduce', c=C, dtype=dtype, n_pixels=npix, x=xg, sums=old)
    torch.testing.assert_close(old.cpu(), want0, **tol)

    # fused statistics + finalisation == finalize on the same sums
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.2).to(DEV)
    outs = []
    for fused in (True, False):
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        cnt = torch.zeros((), dtype=torch.int64, device=DEV)
        m, r, sc, sh = (torch.empty(C, device=DEV) for _ in range(4))
        sums = torch.empty(2 * C, dtype=torch.float64, device=DEV)
        kw = dict(c=C, eps=1e-5, momentum=0.1, gamma=gamma, beta=beta, mean=m, rstd=r, scale=sc, shift=sh, running_mean=rm,
                  running_var=rv, counter=cnt)
        if fused:
            ops.bn_op('stats', dtype=dtype, n_pixels=npix, x=xg, sums=sums, ws=ws, **kw)
        else:
            ops.bn_op('reduce', c=C, dtype=dtype, n_pixels=npix, x=xg, sums=sums, ws=ws)
            ops.bn_op('finalize', count=npix, sums=sums, **kw)
        outs.append([t.cpu() for t in (m, r, sc, sh, rm, rv, cnt, sums)])
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    assert int(outs[0][6]) == 1
    torch.testing.assert_close(outs[0][0].double(), want0[:C] / npix, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('C,dtype', [(64, torch.float32), (256, torch.bfloat16), (2208, torch.float32)])
def test_sample_groups_equal_separate_passes(C, dtype):
    """batch_norm_act(groups=3) over [a; b; c] == three calls in that order: outputs, running statistics (moved once per group,
    in order) and every gradient (the parameter gradients of the groups add up)."""
    from cutmix_semisup_seg_amd import ops
    g = torch.Generator().manual_seed(C)
    G, N, H, W = 3, 2, 9, 7
    x = (torch.randn(G * N, H, W, C, generator=g) * torch.tensor([1.0, 2.0, 0.5]).repeat_interleave(N).view(-1, 1, 1, 1) + 0.2).to(dtype)
    res = torch.randn(G * N, H, W, C, generator=g).to(dtype)
    dy = torch.randn(G * N, H, W, C, generator=g).to(dtype)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2

    def run(groups):
        xg, rg = x.to(DEV).requires_grad_(True), res.to(DEV).requires_grad_(True)
        gg, bg = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        if groups == 1:
            ys = [ops.batch_norm_act(xg[i * N:(i + 1) * N], gg, bg, rm, rv, 0.1, 1e-5, relu=True, res=rg[i * N:(i + 1) * N])
                  for i in range(G)]
            y = torch.cat(ys, 0)
        else:
            y = ops.batch_norm_act(xg, gg, bg, rm, rv, 0.1, 1e-5, relu=True, res=rg, groups=G)
        y.backward(dy.to(DEV))
        return [t.detach().float().cpu() for t in (y, rm, rv, xg.grad, rg.grad, gg.grad, bg.grad)]

    a, b = run(1), run(G)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for name, u, v in zip(('y', 'running_mean', 'running_var', 'dx', 'dres', 'dgamma', 'dbeta'), a, b):
        torch.testing.assert_close(u, v, rtol=tol, atol=tol, msg=lambda m, name=name: name + ': ' + m)


@pytest.mark.parametrize('fenced', [0, 1], ids=['write_through', 'release_acquire_fences'])
def test_last_block_protocol_never_reads_a_stale_partial(fenced, monkeypatch):
    """The reductions pass their partial sums through write-through stores and sc1 loads without fences (csrc/bn.hip): launches
    alternating between two inputs on ONE workspace, with convolutions on another stream keeping the L2s busy, must reproduce
    each input's first result bit for bit (a stale partial would be the other input's). tools/bn_stress.py is the long version.
    CMS_BN_FENCE=1 (read at every launch) adds the release / acquire fences of the HIP memory model: the fallback a user can
    switch on, held to the same test -- and to the same bits as the default."""
    from cutmix_semisup_seg_amd import ops
    monkeypatch.setenv('CMS_BN_FENCE', str(fenced))
    torch.manual_seed(0)
    P, C, G = 16810, 1024, 2
    xs = [(torch.randn(P, C, device=DEV) * (1.0 + i) + i).bfloat16() for i in range(2)]
    ws = ops.bn_workspace(P, C, DEV, G)
    side = torch.cuda.Stream()
    xc = torch.randn(8, 41, 41, 256, device=DEV).bfloat16()
    wc = (torch.randn(9, 256, 256, device=DEV) * 0.05).bfloat16()
    taps = ops.conv_taps(3, 3, 2, 2)
    ref, outs = {}, []
    for it in range(400):
        if it % 40 == 0:
            with torch.cuda.stream(side):
                for _ in range(6):
                    ops.conv_igemm(xc, wc, taps)
        out = torch.empty(G * 2 * C, dtype=torch.float64, device=DEV)
        ops.bn_op('reduce', c=C, dtype=torch.bfloat16, n_pixels=P, groups=G, x=xs[it & 1], sums=out, ws=ws)
        outs.append((it & 1, out))
    torch.cuda.synchronize()
    for i, o in outs:
        assert torch.equal(ref.setdefault(i, o), o)
    assert not torch.equal(ref[0], ref[1])


def _sync_groups_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        q.put((rank,) + _grouped_bn_pass([rank, 2 + rank]))
    finally:
        dist.destroy_process_group()


def _grouped_bn_pass(idx):
    """batch_norm_act with TWO sample groups on the samples `idx` of a fixed 4-sample batch [s0 s1 | s2 s3]."""
    from cutmix_semisup_seg_amd import ops
    dev = torch.device(DEV)
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(6)
    C = 64
    x = torch.randn(4, 9, 7, C, generator=g) * torch.tensor([1.0, 1.5, 0.5, 2.0]).view(4, 1, 1, 1)
    dy = torch.randn(4, 9, 7, C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    xs = x[idx].contiguous().to(dev).requires_grad_(True)
    gg, bg = gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y = ops.batch_norm_act(xs, gg, bg, rm, rv, 0.1, 1e-5, relu=True, groups=2)
    y.backward(dy[idx].contiguous().to(dev))
    return (y.detach().cpu().numpy(), xs.grad.cpu().numpy(), gg.grad.cpu().numpy(), bg.grad.cpu().numpy(), rm.cpu().numpy(),
            rv.cpu().numpy())


def test_sync_batchnorm_with_sample_groups_two_ranks_equal_one_process():
    """Round 4: sample groups under data parallelism (the DeepLab v3+ head of BASELINE configs[3] on 4 GPUs): each rank holds one
    sample of each group; the per-group sums are all-reduced in one exchange and finalised group by group."""
    import torch.multiprocessing as mp
    want = _grouped_bn_pass([0, 1, 2, 3])
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_groups_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        np.testing.assert_allclose(res[r][1], want[0][[r, 2 + r]], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(res[r][2], want[1][[r, 2 + r]], rtol=5e-4, atol=5e-5)
        np.testing.assert_allclose(res[r][5], want[4], rtol=1e-5, atol=1e-6)       # running mean: global, both groups in order
        np.testing.assert_allclose(res[r][6], want[5], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(res[0][3] + res[1][3], want[2], rtol=5e-4, atol=5e-5)     # local dgamma / dbeta add up
    np.testing.assert_allclose(res[0][4] + res[1][4], want[3], rtol=5e-4, atol=5e-5)


@pytest.mark.parametrize('G', [1, 2])
@pytest.mark.parametrize('C,dtype', [(64, torch.bfloat16), (256, torch.bfloat16), (2048, torch.bfloat16), (72, torch.float32)])
@pytest.mark.parametrize('with_res', [False, True])
def test_relu_mask_bits_replace_y_in_the_backward_passes_bit_for_bit(C, dtype, with_res, G):
    """Round 5: the normalising launch writes [stored y > 0] as bits ([pixel rows][C / 8], bit e = channel 8 v + e) and the two
    backward passes read those instead of y -- same sums, same dx / dres, to the bit."""
    from cutmix_semisup_seg_amd import ops
    g = torch.Generator().manual_seed(C + G)
    N, H, W = 2 * G, 19, 23
    P = N * H * W
    u = (torch.randn(N, H, W, C, generator=g) * 1.3).to(dtype).to(DEV)
    res = (torch.randn(N, H, W, C, generator=g) * 0.7).to(dtype).to(DEV) if with_res else None
    dy = torch.randn(N, H, W, C, generator=g).to(dtype).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.2).to(DEV)
    new = lambda n: torch.empty(n, dtype=torch.float32, device=DEV)
    mean, rstd, scale, shift = new(G * C), new(G * C), new(G * C), new(G * C)
    ws = ops.bn_workspace(P, C, DEV, G)
    ops.bn_op('stats', c=C, dtype=dtype, n_pixels=P, groups=G, eps=1e-5, momentum=0.1, x=u, ws=ws, gamma=gamma, beta=beta,
              mean=mean, rstd=rstd, scale=scale, shift=shift)
    y, y2 = torch.empty_like(u), torch.empty_like(u)
    bits = torch.zeros(P * C // 8, dtype=torch.uint8, device=DEV)
    ops.bn_op('apply', c=C, dtype=dtype, n_pixels=P, groups=G, relu=True, x=u, res=res, y=y, scale=scale, shift=shift)
    ops.bn_op('apply', c=C, dtype=dtype, n_pixels=P, groups=G, relu=True, x=u, res=res, y=y2, scale=scale, shift=shift,
              mask_bits=bits)
    assert torch.equal(y, y2)
    want_bits = ((y.view(P, C // 8, 8) > 0).to(torch.int32) << torch.arange(8, device=DEV, dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(bits.view(P, C // 8), want_bits)
    assert 0.2 < float((y > 0).float().mean()) < 0.8
    out = {}
    for key, kw in (('y', dict(y=y)), ('bits', dict(mask_bits=bits))):
        sums = torch.empty(G * 2 * C, dtype=torch.float64, device=DEV)
        ops.bn_op('reduce_bwd', c=C, dtype=dtype, n_pixels=P, groups=G, x=u, dy=dy, mean=mean, rstd=rstd, sums=sums, ws=ws, **kw)
        dx, dres = torch.empty_like(u), torch.empty_like(u)
        ops.bn_op('bwd_apply', c=C, dtype=dtype, n_pixels=P, groups=G, count=P // G, x=u, dy=dy, dx=dx, dres=dres, mean=mean,
                  rstd=rstd, gamma=gamma, sums=sums, **kw)
        torch.cuda.synchronize()
        out[key] = (sums.clone(), dx, dres)
    for a, b in zip(out['y'], out['bits']):
        assert torch.equal(a, b)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32], ids=['bf16', 'fp32'])
@pytest.mark.parametrize('relu,with_res', [(True, True), (True, False), (False, False), (False, True)])
def test_frozen_bn_act_one_launch_vs_tensor_ops(dtype, relu, with_res):
    """(round 6) ops.frozen_bn_act -- an eval-mode BatchNorm with a non-trainable affine (+ residual, + ReLU) as one launch forward and
    one backward (cms_bn_apply / cms_frozen_bn_act_bwd) -- against the fp64 formula: output, gradient wrt the input and the residual."""
    from cutmix_semisup_seg_amd import ops
    g = torch.Generator(device=DEV).manual_seed(11)
    n, h, w, c = 3, 17, 19, 64
    x = torch.randn(n, h, w, c, generator=g, device=DEV).to(dtype).requires_grad_(True)
    res = torch.randn(n, h, w, c, generator=g, device=DEV).to(dtype).requires_grad_(True) if with_res else None
    scale = torch.rand(c, generator=g, device=DEV) + 0.5
    shift = torch.randn(c, generator=g, device=DEV)
    dy = torch.randn(n, h, w, c, generator=g, device=DEV).to(dtype)
    y = ops.frozen_bn_act(x, scale, shift, relu=relu, res=res)
    y.backward(dy)
    xd = x.detach().double()
    v = xd * scale.double() + shift.double() + (res.detach().double() if with_res else 0.0)
    want = v.clamp_min(0.0) if relu else v
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert float((y.double() - want).abs().max()) <= tol * (1.0 + float(want.abs().max()))
    mask = (y.detach() > 0).double() if relu else torch.ones_like(want)        # (the kernel masks with the STORED output)
    dyd = dy.double() * mask
    assert float((x.grad.double() - dyd * scale.double()).abs().max()) <= tol * (1.0 + float(dyd.abs().max()) * 1.5)
    if with_res:
        assert float((res.grad.double() - dyd).abs().max()) <= tol * (1.0 + float(dyd.abs().max()))
