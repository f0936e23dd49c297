"""
Data-parallel smoke test of the CutMix mean-teacher step over RCCL, for a box with >= 2 GPUs.

    torchrun --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/rccl_smoke.py
    (or: python tools/rccl_smoke.py --gpus 2        -- re-executes itself under torch.distributed.run)

One process per GPU, backend "nccl" (= RCCL on ROCm), HSA_ENABLE_IPC_MODE_LEGACY=0. Every rank builds the same seeded
ResNet-[1,1,1,1] DeepLab v2 student / teacher, draws ITS OWN supervised / unsupervised shards and box masks, and runs 3
iterations of the fused-batch step (bucketed gradient all-reduce from the weight-gradient stream, 16-byte confidence
all-reduce). Asserted: the RCCL group really spans N ranks; the all-reduced gradient arena equals the sum of the ranks'
local gradients (iteration 1, recomputed without the exchange); after 3 steps the student AND teacher arenas are bit-identical
on every rank; per-bucket (bytes, issue-to-wait ms) are printed. Exit code 0 = all of it held.

`--backend gloo --same_device` runs the same logic with every rank on cuda:0 (a 1-GPU box: what the build could run).
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=0, help='re-execute under torch.distributed.run with this many ranks')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'])
    ap.add_argument('--same_device', action='store_true', help='every rank on cuda:0 (gloo only)')
    ap.add_argument('--allreduce_dtype', default='fp32', choices=['fp32', 'bf16'])
    ap.add_argument('--arch', default='deeplab2', choices=['deeplab2', 'resnet101_deeplabv3plus_imagenet'],
                    help='deeplab2 = the tiny ResNet-[1,1,1,1] DeepLab v2; the v3+ name builds the registry network (its head keeps '
                         'batch-statistics BatchNorm even under --freeze_bn: SyncBN with sample groups, BASELINE configs[3])')
    ap.add_argument('--no_freeze_bn', action='store_true',
                    help='batch-statistics BatchNorm everywhere (the reference CLI default): the executor passes all-reduce '
                         'their per-group sums inside the recorded programs')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        import socket
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + \
              [a for a in sys.argv[1:] if a not in ('--gpus', str(args.gpus))]
        os.execvp(cmd[0], cmd)

    import numpy as np
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    local = 0 if args.same_device else int(os.environ.get('LOCAL_RANK', '0'))
    if args.backend == 'nccl' and args.same_device:
        raise SystemExit('RCCL needs one device per rank')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group(args.backend)
    probe = torch.ones(1, device=dev)
    dist.all_reduce(probe)
    assert int(probe.item()) == world, 'the process group spans {} ranks, expected {}'.format(int(probe.item()), world)

    from cutmix_semisup_seg_amd import ops, optim as fo
    print('rank {}: stream probe after the first collective: {}'.format(rank, ops.probe_streams(dev, again=True)), flush=True)
    from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig, UnsupBatch
    from architectures import deeplab2
    import mask_gen
    import optim_weight_ema

    C, layers, N, H, W = 5, [1, 1, 1, 1], 2, 65, 65

    def build(allreduce):
        torch.manual_seed(7)                                   # identical replicas
        if args.arch == 'deeplab2':
            mk = lambda: deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3)).to(dev)
        else:
            from architectures import network_architectures
            mk = lambda: network_architectures.seg.get(args.arch)(C, pretrained=False).to(dev)
        stu, tea = mk(), mk()
        if args.arch != 'deeplab2':
            # the 'auto' engine leaves the pooled branch's 1 x 1-map convolution to the library, whose result varies in the last
            # bits from launch to launch; bf16 storage amplifies that to ~1e-2 of the gradients (measured: run-to-run, ONE
            # process) -- the sum check below needs the reproducible all-hand-written engine
            stu.engine_kind = tea.engine_kind = 'hip'
        opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=1e-4),
                                 dict(params=list(stu.new_parameters()), lr=1e-3)])
        for p in tea.parameters():
            p.requires_grad = False
        ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
        ema.fuse_into(opt)
        stu.train(); tea.train()
        if not args.no_freeze_bn:
            stu.freeze_batchnorm(); tea.freeze_batchnorm()
        cfg = StepConfig(conf_thresh=0.3, allreduce_dtype=args.allreduce_dtype, deterministic=True)
        step = CutMixMeanTeacherStep(stu, tea, opt, ema, cfg)
        if not allreduce:
            step.world = 1                                     # local gradients only (the yardstick of the sum check)
        step.time_buckets = allreduce
        return stu, tea, opt, step

    def batches(it):
        g = torch.Generator(device=dev).manual_seed(1000 * rank + it)
        im = lambda: torch.randn(N, 3, H, W, generator=g, device=dev).bfloat16()
        y = torch.randint(0, C, (N, 1, H, W), generator=g, device=dev).to(torch.uint8)
        r = mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(N, (H, W), rng=np.random.RandomState(50 * rank + it))
        return im(), y, [UnsupBatch(im(), ops.ranges_to_device(r, dev), x1_tea=im())]

    # (1) local gradients of iteration 0 (no exchange) -> their sum over ranks, computed by a plain all-reduce
    stu0, _, opt0, step0 = build(False)
    torch.manual_seed(4242 + rank)                             # (dropout of the DeepLab v3+ head: the same draw in both runs)
    step0(*batches(0))
    want = opt0.arena.grad.clone()
    dist.all_reduce(want)
    # (2) the data-parallel run
    stu, tea, opt, step = build(True)
    for it in range(3):
        torch.manual_seed(4242 + rank + 100 * it)
        res = step(*batches(it))
        if it == 0:
            got = opt.arena.grad.clone()
            err = float((got - want).abs().max() / (want.abs().max() + 1e-30))
            tol = 5e-6 if args.allreduce_dtype == 'fp32' else 2e-2          # (fp32: summation order of the exchange)
            if err > tol and rank == 0:                                      # name the tensors that differ
                worst = []
                for seg in opt.arena.segments:
                    if not seg.requires_grad:
                        continue
                    a, b = opt.arena.view(seg.key, got), opt.arena.view(seg.key, want)
                    worst.append((float((a - b).abs().max() / (want.abs().max() + 1e-30)), seg.key))
                for e, name in sorted(worst, reverse=True)[:8]:
                    print('   {:.3e}  {}'.format(e, name))
            assert err <= tol, 'all-reduced gradients differ from the sum of the local ones: {:.3e}'.format(err)
    torch.cuda.synchronize()
    timing = step.bucket_timing()
    # (3) identical replicas (with batch statistics: SyncBN moved the running statistics identically on every rank)
    if args.no_freeze_bn or args.arch != 'deeplab2':
        ex = getattr(stu, '_hip_executor', None)
        on_executor = ex is not None and any(p.host_ops for p in ex.programs())
        if rank == 0:
            print('batch-statistics passes: executor programs carry SyncBN all-reduces = {}; sample groups = {}'.format(
                on_executor, step._sample_groups(N, batches(0)[2], True)))
    for name, net in (('student', stu), ('teacher', tea)):
        flat = net._cms_arena.flat
        ref = flat.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(flat, ref), '{} replicas diverged on rank {}'.format(name, rank)
    vals = {k: float(v) for k, v in res.items()}
    if rank == 0:
        print('rccl_smoke OK: backend {} world {} rccl_world_size {} allreduce {} last losses {} buckets {}'.format(
            args.backend, world, int(probe.item()), args.allreduce_dtype, vals, timing))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
