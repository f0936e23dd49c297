"""
GPU, two processes sharing cuda:0 over gloo: the data-parallel path of the step as the product runs it -- gradient
buckets all-reduced from the weight-gradient stream while the backward pass continues (step.GradBuckets), the
confidence / CE statistics exchange, identical replicas after the optimizer step. (RCCL needs one device per rank; the
protocol, the hooks and the stream ordering are backend-independent. The 8-GPU bench itself is run by the driver.)
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dev, C, layers):
    from architectures import deeplab2
    from cutmix_semisup_seg_amd import optim as fo
    import optim_weight_ema
    torch.manual_seed(77)
    stu = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3)).to(dev)
    tea = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3)).to(dev)
    with torch.no_grad():
        for m in stu.modules():                       # healthier gradients than N(0, 0.01)
            if isinstance(m, torch.nn.Conv2d):
                m.weight.normal_(0, (1.0 / (m.in_channels * m.kernel_size[0] * m.kernel_size[1])) ** 0.5)
    opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=1e-4),
                             dict(params=list(stu.new_parameters()), lr=1e-3)])
    for p in tea.parameters():
        p.requires_grad = False
    ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
    ema.fuse_into(opt)
    stu.train(); tea.train(); stu.freeze_batchnorm(); tea.freeze_batchnorm()
    return stu, tea, opt, ema


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cutmix_semisup_seg_amd import ops
        from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig, UnsupBatch
        import mask_gen
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)
        C, layers, N, H, W = 5, (1, 1, 2, 1), 2, 65, 65
        g = torch.Generator(device=dev).manual_seed(1000 + rank)          # every rank its own shard
        im = lambda: torch.randn(N, 3, H, W, generator=g, device=dev).bfloat16()
        x, x0, x1 = im(), im(), im()
        y = torch.randint(0, C, (N, 1, H, W), generator=g, device=dev).to(torch.uint8)
        ranges = ops.ranges_to_device(mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(
            N, (H, W), rng=np.random.RandomState(rank)), dev)
        ub = UnsupBatch(x0, ranges, x1_tea=x1)

        # (1) local gradients of this rank's shard: same model, the collective paths switched off
        stu, tea, opt, ema = _build(dev, C, layers)
        step = CutMixMeanTeacherStep(stu, tea, opt, ema, StepConfig(conf_thresh=0.0))
        step.world = 1
        step(x, y, [ub])
        local = opt.arena.grad.clone()

        # (2) the distributed step: bucketed all-reduce from the weight-gradient stream
        stu, tea, opt, ema = _build(dev, C, layers)
        step = CutMixMeanTeacherStep(stu, tea, opt, ema, StepConfig(conf_thresh=0.0))
        assert step.world == world
        step(x, y, [ub])
        torch.cuda.synchronize()
        assert step._bucket_obj is not None and step._bucket_obj.hi == 0, 'bucketed path not taken'
        assert opt.grad_scale == 1.0 / world
        reduced = opt.arena.grad.clone()
        total = local.clone()
        dist.all_reduce(total)                                              # what the buckets must add up to
        w = stu.state_dict()['layer3.1.conv2.weight'].float().cpu()
        tw = tea.state_dict()['layer3.1.conv2.weight'].float().cpu()
        q.put((rank, reduced.cpu().numpy(), total.cpu().numpy(), local.cpu().numpy(), w.numpy(), tw.numpy()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_bucketed_allreduce_and_identical_replicas():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, red0, tot0, loc0, w0, tw0), (_, red1, tot1, loc1, w1, tw1) = res
    assert np.abs(loc0 - loc1).max() > 0                       # the shards really differ
    # every element reduced exactly once: reduced == sum of the ranks' local gradients (fp32 atomics in the weight
    # gradient make two runs of the same backward differ in the last bits, hence a tolerance)
    scale = np.abs(tot0).max()
    assert np.abs(red0 - tot0).max() <= 2e-3 * scale
    assert np.abs(red1 - tot1).max() <= 2e-3 * scale
    np.testing.assert_array_equal(red0, red1)                  # both ranks hold the same reduced buffer ...
    np.testing.assert_array_equal(w0, w1)                      # ... and stay identical replicas after Adam + EMA
    np.testing.assert_array_equal(tw0, tw1)


def test_rccl_single_rank_exchange_paths():
    """(round 6) RCCL itself, on one GPU: backend "nccl" with world size 1 in a fresh process (tools/rccl_single_rank.py) -- the
    stream probe after the first collective, GradBuckets driven from a side stream exactly like the fused step (fp32 and through
    the bf16 staging arena; late writers, an immediate reader: a missing stream edge shows as stale data), the small statistics
    all-reduces between launches, three iterations of the fused step with RCCL resident. The gloo tests above cannot see
    stream-ordering mistakes (gloo's collectives on device tensors synchronise the host)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('LOCAL_RANK', 'GROUP_RANK', 'LOCAL_WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'rccl_single_rank.py')], env=env, capture_output=True, text=True,
                       timeout=600)
    if r.returncode == 77:
        pytest.skip('RCCL did not come up on this box: ' + r.stdout.strip()[-300:])
    # (RCCL prints its version banner through C stdio when the process ends: 'OK' is a line of the output, not its last one)
    assert r.returncode == 0 and 'OK' in r.stdout.splitlines(), (r.stdout[-2000:], r.stderr[-2000:])
