#!/usr/bin/env python
"""
bench.py -- train images/sec of the CutMix mean-teacher step (student + teacher), BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank / GPU)

A "step" is one full iteration of train_seg_semisup_mask_mt.py:287-476 on synthetic inputs already resident in HBM:
student fwd/bwd on the supervised batch, 2 teacher forwards, student fwd/bwd on the CutMix-mixed batch, fused masked
consistency + CE losses, fused Adam + EMA, (N > 1) one RCCL all-reduce of the flat gradient arena.
Workload at N = 1: BASELINE configs[1] -- DeepLab v2 / ResNet-101, 10 x 3 x 321 x 321, 21 classes, CutMix, bf16.
`--workload cityscapes` selects configs[2] (4 x 3 x 512 x 1024 per GPU, 19 classes, paired colour-aug layout).
Weak scaling: the per-GPU batch is fixed. Prints ONE JSON line on rank 0.

Extra objects in the line
  roofline      the dominant hand-written kernel of the step (see --roofline_kernel), timed live with HIP events on
                the launch stream inside the timed region; algorithmic bytes per launch from DESIGN.md.
  cpu_baseline  the CPU oracle's restatement of the same step (kind "port"), timed on this box's host cores on a
                bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA ~2.5 PFLOP/s

WORKLOADS = {
    'pascal': dict(name='deeplab2-resnet101 cutmix mean-teacher step, 10x3x321x321, 21 classes (BASELINE configs[1])',
                   batch=10, H=321, W=321, classes=21, paired=False),
    'pascal_v3plus': dict(name='deeplab3plus-resnet101 cutmix mean-teacher step, 10x3x513x513, 21 classes '
                               '(BASELINE configs[3]); library convolutions, batch-statistics head => separate passes',
                          batch=10, H=513, W=513, classes=21, paired=False, arch='resnet101_deeplabv3plus_imagenet'),
    'cityscapes': dict(name='deeplab2-resnet101 cutmix mean-teacher step, 4x3x512x1024 per GPU, 19 classes, '
                            'paired colour-aug layout (BASELINE configs[2])',
                       batch=4, H=512, W=1024, classes=19, paired=True),
}


def cpu_baseline(workload, seconds_budget=30.0):
    """Oracle step on the host cores, bounded sample: batch 2 (the GPU run uses the full batch), >= 2 timed iters."""
    import numpy as np
    import torch
    from oracle import deeplab2 as odl, step as ostep, boxmask as obox
    torch.manual_seed(0)
    # conv-heavy fp32 work stops scaling (and regresses) long before 100+ threads; use at most 32
    cores = min(torch.get_num_threads(), 32)
    torch.set_num_threads(cores)
    C, H, W = workload['classes'], workload['H'], workload['W']
    N = 2 if H * W <= 321 * 321 else 1
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, 3, H, W, generator=g)
    y = torch.randint(0, C, (N, 1, H, W), generator=g)
    ux0, ux1 = torch.randn(N, 3, H, W, generator=g), torch.randn(N, 3, H, W, generator=g)
    ones = torch.ones(N, 1, H, W)
    m = torch.tensor(obox.generate_params(N, (H, W), 0.5, invert=True, rng=np.random.RandomState(0)).astype(np.float32))
    if 'v3plus' in workload.get('arch', ''):
        from oracle import deeplab3plus as o3, step_v3plus as sv
        N = 2                                   # batch statistics in the head need more than one sample
        x, y, ux0, ux1, ones, m = (torch.cat([t, t.flip(3)], 0) for t in (x, y, ux0, ux1, ones, m))
        S = sv.StepStateV3Plus(o3.closed_form_state(C), C, lr=3e-5)
        run = lambda: sv.train_iteration(S, x, y, ux0, ux1, ones, ones, m)
        what, min_iters = 'oracle/step_v3plus.py', 1
    else:
        S = ostep.StepState(odl.closed_form_state(C), C, opt='adam', lr=3e-5)
        run = lambda: ostep.train_iteration(S, x, y, ux0, ux1, ones, ones, m)
        what, min_iters = 'oracle/step.py', 2
    run()                                                             # warm-up
    times = []
    t_start = time.time()
    while len(times) < min_iters or (time.time() - t_start < seconds_budget * 0.6 and len(times) < 8):
        t0 = time.time()
        run()
        times.append(time.time() - t0)
    t = sum(times) / len(times)
    return dict(value=N / t, unit='images/sec', cores=cores, kind='port',
                sample=what + ' (PyTorch-CPU fp32 restatement of the reference step), batch {} of {}x{} '
                       '(GPU run: batch {}), {} timed iterations after 1 warm-up, {:.2f} s/iter'.format(
                           N, H, W, workload['batch'], len(times), t))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='pascal')
    ap.add_argument('--dtype', choices=['bf16', 'fp32'], default='bf16')
    ap.add_argument('--roofline_kernel', choices=['conv', 'adam_ema', 'consistency'], default=None,
                    help='default: conv (the MFMA convolution) for the DeepLab v2 workloads, adam_ema otherwise')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_fuse_batches', action='store_true')
    ap.add_argument('--conv_tile', type=int, default=0, help='experiment: force a conv tile code (256, 1128, 128)')
    ap.add_argument('--tile_rule', default='', help='experiment: cout:tile[,cout:tile...] per-layer tile codes')
    ap.add_argument('--no_overlap', action='store_true', help='single stream: no teacher / weight-gradient overlap')
    ap.add_argument('--no_roofline_events', action='store_true', help='skip the per-launch event brackets')
    ap.add_argument('--roofline_sample', type=int, default=5,
                    help='bracket every k-th launch of the conv kernel with events (1 = all; the brackets cost ~3 %% '
                         'of the step when every launch carries them)')
    args = ap.parse_args()
    args.roofline_sample = max(1, args.roofline_sample)
    if args.roofline_kernel is None:
        args.roofline_kernel = 'conv' if 'arch' not in WORKLOADS[args.workload] else 'adam_ema'
    args.roofline_sample_used = args.roofline_sample

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus {} but WORLD_SIZE={}'.format(args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl')

    from cutmix_semisup_seg_amd import ops, optim as fo
    from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig, UnsupBatch
    from architectures import network_architectures
    import mask_gen
    import optim_weight_ema

    wl = WORKLOADS[args.workload]
    B, H, W, C = wl['batch'], wl['H'], wl['W'], wl['classes']
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32

    torch.manual_seed(12345)                       # identical replicas on every rank
    Net = network_architectures.seg.get(wl.get('arch', 'resnet101_deeplab_imagenet'))
    stu, tea = Net(C, pretrained=False).to(dev), Net(C, pretrained=False).to(dev)
    stu.compute_dtype = tea.compute_dtype = dtype
    lr = 3e-5                                      # run_pascal_aug_experiments.sh
    opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=lr * 0.1),
                             dict(params=list(stu.new_parameters()), lr=lr)])
    for p in tea.parameters():
        p.requires_grad = False
    ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
    ema.fuse_into(opt)
    stu.train(); tea.train(); stu.freeze_batchnorm(); tea.freeze_batchnorm()      # --freeze_bn
    cfg = StepConfig(mask_mode='mix', cons_loss_fn='var', cons_weight=1.0, conf_thresh=0.97, conf_per_pixel=False,
                     fuse_batches=not args.no_fuse_batches, compute_dtype=dtype,
                     overlap_teacher=not args.no_overlap)
    step = CutMixMeanTeacherStep(stu, tea, opt, ema, cfg)
    if args.no_overlap and hasattr(stu, 'hip_executor') and dtype == torch.bfloat16:
        stu.hip_executor().overlap_wgrad = False
    if args.conv_tile and hasattr(stu, 'hip_executor'):
        stu.hip_executor().conv_tile = tea.hip_executor().conv_tile = args.conv_tile
    if args.tile_rule and hasattr(stu, 'hip_executor'):
        rules = {int(a): int(b) for a, b in (kv.split(':') for kv in args.tile_rule.split(','))}
        stu.hip_executor().tile_rules = tea.hip_executor().tile_rules = rules

    gen = torch.Generator(device=dev).manual_seed(12345 + rank)
    mask_rng = np.random.RandomState(12345 + rank)
    boxgen = mask_gen.BoxMaskGenerator(0.5, invert=True)

    def images():
        return torch.randn(B, 3, H, W, generator=gen, device=dev).to(dtype)

    # a small pool of resident synthetic batches, cycled (inputs are in HBM before the timed region starts)
    pool = []
    for _ in range(2):
        y = torch.randint(0, C, (B, 1, H, W), generator=gen, device=dev)
        y[torch.rand(B, 1, H, W, generator=gen, device=dev) < 0.05] = 255
        pool.append(dict(x=images(), y=y.to(torch.uint8), x0=images(), x1=images(),
                         x0s=images() if wl['paired'] else None, x1s=images() if wl['paired'] else None))

    # roofline instrumentation: bracket every launch of the chosen kernel with events on the launch stream
    ev_pairs = []
    timing_on = [False]
    roof = dict(bound='hbm', peak=HBM_PEAK_GBS, unit='GB/s')
    work_per_launch = []            # algorithmic bytes or FLOPs of every timed launch
    step_flops = [0.0]              # algorithmic MFMA FLOPs of ALL convolution launches in the timed region
    launch_no = [0]
    conv_bytes = [0.0]              # algorithmic HBM bytes of the same launches (operands once, output once)
    if args.roofline_kernel == 'conv':
        # the dominant kernel of the step: conv_igemm_kernel (csrc/conv.hip) -- every forward and data-gradient
        # convolution of the backbone. Algorithmic FLOPs of a launch = 2 * pixels * Cout * Cin * taps.
        orig_conv = ops.conv_igemm

        def conv_flops(x, w_packed, k):
            ohw = k.get('out_hw')
            npix = x.shape[0] * (ohw[0] * ohw[1] if ohw is not None else x.shape[1] * x.shape[2])
            return 2.0 * npix * w_packed.shape[1] * w_packed.shape[2] * w_packed.shape[0]

        def timed_conv(x, w_packed, taps, *a, **k):
            if not timing_on[0]:
                return orig_conv(x, w_packed, taps, *a, **k)
            fl = conv_flops(x, w_packed, k)
            step_flops[0] += fl
            npix_out = fl / (2.0 * w_packed.shape[1] * w_packed.shape[2] * w_packed.shape[0])
            conv_bytes[0] += 2.0 * (x.numel() + w_packed.numel()) + npix_out * w_packed.shape[1] * (
                (4.0 if k.get('out_f32_nchw') is not None else 2.0) + (2.0 if k.get('res') is not None else 0.0)
                + (2.0 if k.get('mask_src') is not None else 0.0))
            launch_no[0] += 1
            if launch_no[0] % args.roofline_sample:
                return orig_conv(x, w_packed, taps, *a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig_conv(x, w_packed, taps, *a, **k)
            e1.record()
            work_per_launch.append(fl)
            ev_pairs.append((e0, e1))
            return r

        orig_wgrad = ops.conv_wgrad

        def counted_wgrad(du, x, taps, dw, *a, **k):
            if timing_on[0]:
                step_flops[0] += 2.0 * du.shape[0] * du.shape[1] * du.shape[2] * du.shape[3] * x.shape[3] * len(taps)
            return orig_wgrad(du, x, taps, dw, *a, **k)
        ops.conv_wgrad = counted_wgrad
        ops.conv_igemm = timed_conv
        roof = dict(bound='mfma', peak=MFMA_PEAK_TFLOPS, unit='TFLOP/s')
        kname = 'conv_igemm_kernel (MFMA implicit-GEMM convolution, forward + data-gradient launches of the backbone)'
    elif args.roofline_kernel == 'adam_ema':
        from cutmix_semisup_seg_amd import _lib
        orig_launch = _lib.fn['cms_adam_ema_step']          # optim.py launches through this table

        def timed_launch(desc, stream):
            if not timing_on[0]:
                return orig_launch(desc, stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = orig_launch(desc, stream)
            e1.record()
            ev_pairs.append((e0, e1))
            work_per_launch.append(opt.arena.total * 40.0)   # 5 fp32 reads + 4 fp32 writes + 2 bf16 copies
            return rc
        _lib.fn['cms_adam_ema_step'] = timed_launch
        kname = 'optim_ema_kernel<ADAM> (fused Adam + teacher EMA over the 44.2M-element arena)'
    else:
        orig_fwd = ops.consistency_forward

        def timed_fwd(*a, **k):
            if not timing_on[0]:
                return orig_fwd(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig_fwd(*a, **k)
            e1.record()
            ev_pairs.append((e0, e1))
            work_per_launch.append((2 * C + 2) * B * H * W * 4.0)   # reference-equivalent traffic, SURVEY 8(d)
            return r
        ops.consistency_forward = timed_fwd
        kname = 'cons_fwd_kernel (+ second-stage reduce + finalize)'

    def one_step(i):
        b = pool[i % len(pool)]
        ranges = ops.ranges_to_device(boxgen.generate_ranges(B, (H, W), rng=mask_rng), dev)
        ub = UnsupBatch(b['x0'], ranges, x1_tea=b['x1'], x0_stu=b['x0s'], x1_stu=b['x1s'])
        return step(b['x'], b['y'], [ub])

    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timing_on[0] = not args.no_roofline_events
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = one_step(i)
    t_enqueue = time.perf_counter() - t0         # host time to enqueue the K steps (== elapsed when launch-bound)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing_on[0] = False
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)

    last = {k: (None if v is None else float(v)) for k, v in res.items()}
    if not np.isfinite(last['sup_loss']):
        raise SystemExit('non-finite loss in the bench loop: {}'.format(last))

    # Outside the timed region: the same kernel with the GPU to itself (single stream, every launch bracketed). In
    # the timed region the teacher pass and the weight gradients run concurrently on other streams, so a launch's
    # event-to-event time there includes the share of the machine the co-running kernels took.
    isolated = None
    timed_pairs, timed_work, timed_flops = list(ev_pairs), list(work_per_launch), step_flops[0]
    if world == 1 and args.roofline_kernel == 'conv' and not args.no_overlap and not args.no_roofline_events \
            and dtype == torch.bfloat16:
        del ev_pairs[:], work_per_launch[:]
        cfg.overlap_teacher = False
        stu.hip_executor().overlap_wgrad = False
        args.roofline_sample = 1
        timing_on[0] = True
        for i in range(3):
            one_step(i)
        torch.cuda.synchronize()
        timing_on[0] = False
        ms_iso = [a.elapsed_time(b) for a, b in ev_pairs]
        ach = sum(work_per_launch) / (sum(ms_iso) * 1e-3) / 1e12
        isolated = {'achieved': ach, 'frac': ach / roof['peak'], 'avg_launch_ms': float(np.mean(ms_iso)),
                    'launches_timed': len(ms_iso), 'note': 'same kernel, single stream, 3 extra steps after the '
                    'timed region (not part of `value`)'}
    ev_pairs, work_per_launch = timed_pairs, timed_work

    if rank == 0:
        ms_all = [a.elapsed_time(b) for a, b in ev_pairs]
        ms_kernel = float(np.mean(ms_all)) if ms_all else float('nan')
        per_launch = float(np.mean(work_per_launch)) if work_per_launch else float('nan')
        rate = (sum(work_per_launch) / (sum(ms_all) * 1e-3)) if ms_all else float('nan')
        achieved = rate / (1e12 if roof['bound'] == 'mfma' else 1e9)
        out = {
            'metric': 'train images/sec (student+teacher step)',
            'value': args.steps * B * world / elapsed,
            'unit': 'images/sec',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': args.dtype,
            'data': 'synthetic (N(0,1) images, uniform labels with 5% ignore=255, all-ones validity masks, '
                    'seeded box masks; random-init weights)',
            'config': {'workload': wl['name'], 'per_gpu_batch': B, 'global_batch': B * world, 'crop': [H, W],
                       'classes': C, 'parallelism': 'dp{}'.format(world), 'image_forwards_per_sec':
                           4 * args.steps * B * world / elapsed,
                       'fuse_batches': not args.no_fuse_batches, 'stream_overlap': not args.no_overlap,
                       'host_enqueue_ms_per_step': 1e3 * t_enqueue / args.steps,
                       'last_losses': last},
            'roofline': {'bound': roof['bound'], 'kernel': kname, 'achieved': achieved, 'peak': roof['peak'],
                         'unit': roof['unit'], 'frac': achieved / roof['peak'], 'traffic': None,
                         'avg_launch_ms': ms_kernel,
                         'algorithmic_{}_per_launch'.format('flops' if roof['bound'] == 'mfma' else 'bytes'): per_launch,
                         'launches_timed': len(ev_pairs),
                         'sampling': 'every launch' if args.roofline_sample_used == 1 else
                                     'every {}th launch'.format(args.roofline_sample_used)},
        }
        if args.roofline_kernel == 'conv' and launch_no[0]:
            out['roofline']['algorithmic_bytes_per_launch'] = conv_bytes[0] / launch_no[0]
            import glob
            cands = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_pmc_traffic.json')))
            pmc = cands[-1] if cands else ''
            if args.workload == 'pascal' and pmc:
                # HBM bytes per launch from the TCC memory-side counters (separate FETCH_SIZE / WRITE_SIZE passes of
                # this command under rocprofv3, gfx950 correction applied; tools/gpu_pmc_traffic.sh)
                t = json.load(open(pmc))
                out['roofline']['traffic'] = t['traffic_bytes_per_launch']
                out['roofline']['traffic_source'] = 'profiles/{} (rocprofv3 --pmc, bytes per launch)'.format(
                    os.path.basename(pmc))
        if isolated is not None:
            out['roofline']['isolated'] = isolated
        if timed_flops > 0:
            # all MFMA work of the step (forward, data-gradient AND weight-gradient convolutions) over the step time
            out['roofline']['step_mfma'] = {
                'tflop_per_step': timed_flops / args.steps / 1e12,
                'achieved': timed_flops / elapsed / 1e12, 'frac': timed_flops / elapsed / 1e12 / MFMA_PEAK_TFLOPS}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(wl)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
