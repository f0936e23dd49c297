#!/usr/bin/env python
"""
bench.py -- train images/sec of the CutMix mean-teacher step (student + teacher), BASELINE.json's metric
("train images/sec (student+teacher step) at 321x321 and 512x1024, 1/2/4/8 GPU").

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL. The driver launches the ranks through torch.distributed.run; started WITHOUT
WORLD_SIZE in the environment, `--gpus N` re-executes itself under `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1`, so the line always reports the number of ranks that really ran.

A "step" is one full iteration of train_seg_semisup_mask_mt.py:287-476 on synthetic inputs already resident in HBM:
student fwd/bwd on the supervised batch, 2 teacher forwards, student fwd/bwd on the CutMix-mixed batch, fused masked
consistency + CE losses, fused Adam + EMA, (N > 1) bucketed RCCL all-reduce of the flat gradient arena.

Default run = BOTH shapes of the metric, one after the other, in ONE JSON line (rank 0):
  headline `value`            BASELINE configs[1]: DeepLab v2 / ResNet-101, 10 x 3 x 321 x 321, 21 classes, bf16
  `configs[1]` of the line    BASELINE configs[2]: 4 x 3 x 512 x 1024 per GPU, 19 classes, paired colour-aug layout
                              (also as `value_512x1024`); `--workload pascal|cityscapes|pascal_v3plus` runs one only.
Weak scaling: the per-GPU batch is fixed.

Extra objects per workload
  roofline      the dominant hand-written kernels of the step, the cms_conv_igemm launches (conv8_kernel + conv_igemm_*kernel;
                bound: MFMA), timed live with HIP events
                on its launch stream inside the timed region; algorithmic FLOPs per launch = 2 * pixels * Cout * Cin * taps.
  roofline_hbm  the HBM-bound group the north star names -- CutMix paste + masked consistency fwd/bwd + cross entropy
                fwd/bwd + ASPP head convolution -- timed the same way; bytes per SURVEY.md 8(d).
  cpu_baseline  the CPU oracle's restatement of the same step (kind "port"), timed on this box's host cores on a bounded
                sample (rank 0, N = 1 only).
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
MFMA_F32_PEAK_TFLOPS = 157.3   # f32-input MFMA = the fp32 vector rate

WORKLOADS = {
    'pascal': dict(name='deeplab2-resnet101 cutmix mean-teacher step, 10x3x321x321, 21 classes (BASELINE configs[1])',
                   batch=10, H=321, W=321, classes=21, paired=False),
    'pascal_v3plus': dict(name='deeplab3plus-resnet101 cutmix mean-teacher step, 10x3x513x513, 21 classes '
                               '(BASELINE configs[3]); batch-statistics head => separate passes',
                          batch=10, H=513, W=513, classes=21, paired=False, arch='resnet101_deeplabv3plus_imagenet'),
    'cityscapes': dict(name='deeplab2-resnet101 cutmix mean-teacher step, 4x3x512x1024 per GPU, 19 classes, '
                            'paired colour-aug layout (BASELINE configs[2])',
                       batch=4, H=512, W=1024, classes=19, paired=True),
}


LINE_LIMIT = 4096              # the driver keeps a bounded tail of stdout: the final line must stay far below it


def _short(v, n):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + '...'


def _num(v):
    """floats to 6 significant digits (the line is for reading and for the driver's record, the detail file keeps all digits)"""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    if v != v or v in (float('inf'), float('-inf')):
        return None
    return float('%.6g' % v)


def compact_line(full):
    """The ONE line the driver parses: scalars and short objects only, <= LINE_LIMIT bytes, strict JSON (no NaN). `full` is
    the complete result object (what rounds 2-5 printed; now written to bench_detail.json): everything long -- `configs`,
    the `also` bodies, `by_kernel`, `traffic_by_kernel`, `stream_probe`, `roofline_hbm.parts` -- stays there."""
    top = ('metric', 'value', 'unit', 'n_gpus', 'rccl_world_size', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
           'scaling', 'vs_baseline', 'dtype', 'value_321x321', 'value_512x1024')
    out = {k: _num(full[k]) for k in top if k in full}
    out['data'] = 'synthetic'
    cfg = full.get('config') or {}
    out['config'] = {k: _num(_short(v, 100)) for k, v in cfg.items()
                     if not isinstance(v, (dict, list)) and k not in ('parity_config',)}
    if 'parity_config' in cfg:
        out['config']['parity_config'] = ('bf16-storage engine: per-layer teacher-forced parity vs the bf16-storage oracle; the 1e-4 '
                                          'whole-iteration bar is held by --dtype fp32 (DESIGN.md 2.1)'
                                          if full.get('dtype') == 'bf16' else 'fp32 hand-written engine: whole iteration within 1e-4')
    rf = full.get('roofline') or {}
    r = {k: _num(_short(v, 80)) for k, v in rf.items() if not isinstance(v, (dict, list)) and k not in ('sampling', 'traffic_source')}
    tbk = rf.get('traffic_by_kernel') or {}
    for kn, e in tbk.items():                 # the two ratios the review tracks, as scalars
        if 'ratio' in e:
            if 'wgrad8_kernel' in kn:
                r['traffic_ratio_wgrad8'] = _num(float(e['ratio']))
            elif 'conv8_kernel' in kn:
                r['traffic_ratio_conv8'] = _num(float(e['ratio']))
    if rf.get('traffic') and rf.get('algorithmic_bytes_per_launch'):
        r['traffic_ratio'] = _num(float(rf['traffic']) / float(rf['algorithmic_bytes_per_launch']))
    out['roofline'] = r
    if 'roofline_hbm' in full:
        out['roofline_hbm'] = {k: _num(_short(v, 100)) for k, v in full['roofline_hbm'].items()
                               if not isinstance(v, (dict, list)) and k != 'basis'}
    if 'cpu_baseline' in full:
        out['cpu_baseline'] = {k: _num(_short(v, 240)) for k, v in full['cpu_baseline'].items() if not isinstance(v, (dict, list))}
    if 'detail' in full:
        out['detail'] = full['detail']
    line = json.dumps(out, allow_nan=False)
    if len(line) > LINE_LIMIT:                # never again a line the driver cannot keep: drop the optional objects, longest first
        for k in ('roofline_hbm', 'detail'):
            out.pop(k, None)
        out['config'] = {k: v for k, v in out['config'].items() if not isinstance(v, str) or k == 'workload'}
        line = json.dumps(out, allow_nan=False)
    if len(line) > LINE_LIMIT:
        raise RuntimeError('bench line of {} bytes exceeds the {}-byte limit'.format(len(line), LINE_LIMIT))
    return line


def vat_denseunet_run(dev, warmup=6, steps=12):
    """One short run of BASELINE configs[4] (tools/vat_bench.py denseunet): -> an `also` entry."""
    import time
    import torch
    from cutmix_semisup_seg_amd import optim as fo, vat
    from architectures import network_architectures
    import optim_weight_ema
    B, H, W, C = 10, 224, 224, 2
    torch.manual_seed(0)
    Net = network_architectures.seg.get('densenet161unet_imagenet')
    stu, tea = Net(C, pretrained=False).to(dev), Net(C, pretrained=False).to(dev)
    opt = fo.FusedSGD(stu, [dict(params=list(stu.pretrained_parameters()), lr=0.01), dict(params=list(stu.new_parameters()), lr=0.1)],
                      momentum=0.9, nesterov=True, weight_decay=5e-4)
    for p in tea.parameters():
        p.requires_grad = False
    ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
    ema.fuse_into(opt)
    stu.train(); tea.train()
    g = torch.Generator(device=dev).manual_seed(1)
    step = vat.VATMeanTeacherStep(stu, tea, opt, ema, vat.VATConfig(vat_radius=1.0, adaptive_vat_radius=True, cons_loss_fn='kld',
                                                                   cons_weight=0.001, conf_thresh=0.97), generator=g)
    x = torch.randn(B, 3, H, W, generator=g, device=dev).bfloat16()
    xt = torch.randn(B, 3, H, W, generator=g, device=dev).bfloat16()
    y = torch.randint(0, C, (B, 1, H, W), generator=g, device=dev).to(torch.uint8)
    for _ in range(warmup):                       # 2 eager iterations, the capture, the (slower) first replays
        step(x, y, [vat.VATUnsupBatch(xt)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = step(x, y, [vat.VATUnsupBatch(xt)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    captured = sum(1 for v in step._graphs.values() if 'graph' in v)
    del step, opt, ema, stu, tea
    torch.cuda.empty_cache()
    return {'name': 'VAT DenseNet-161 U-Net (BASELINE configs[4])', 'value': B / dt, 'unit': 'images/sec', 'ms_per_step': dt * 1e3,
            'steps': steps, 'warmup': warmup, 'config': {'workload': 'densenet161unet VAT mean-teacher iteration, 10x3x224x224, 2 classes, SGD',
                                                        'hipgraph_replay': bool(captured), 'sup_loss': float(r['sup_loss'])}}


def write_detail(full, path=None):
    """The complete result object next to the script (or under $TMPDIR when the tree is read-only); returns the path or None."""
    import tempfile
    for p_ in ([path] if path else []) + [os.path.join(REPO, 'bench_detail.json'),
                                           os.path.join(tempfile.gettempdir(), 'bench_detail.json')]:
        try:
            with open(p_, 'w') as f:
                json.dump(full, f)
            return p_
        except OSError:
            continue
    return None


def cpu_baseline(workload, seconds_budget=30.0):
    """Oracle step on the host cores, a BOUNDED sample with >= 3 timed iterations after one warm-up: batch 4 of the
    GPU run's 10 at 321 x 321 (~9 s per iteration on 32 cores; the full batch takes ~22 s per iteration and would leave room
    for one), batch 1 at 512 x 1024 (~6 s per image-iteration). images/sec scales with the batch only through better core
    utilisation of the convolutions, which at these sizes is saturated from batch 2 on (0.42-0.45 img/s at batch 10 in round
    3, the same at batch 4)."""
    import numpy as np
    import torch
    from oracle import deeplab2 as odl, step as ostep, boxmask as obox
    torch.manual_seed(0)
    # conv-heavy fp32 work stops scaling (and regresses) long before 100+ threads; use at most 32
    cores = min(torch.get_num_threads(), 32)
    torch.set_num_threads(cores)
    C, H, W = workload['classes'], workload['H'], workload['W']
    N = min(workload['batch'], 4) if H * W <= 321 * 321 else 1
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, 3, H, W, generator=g)
    y = torch.randint(0, C, (N, 1, H, W), generator=g)
    ux0, ux1 = torch.randn(N, 3, H, W, generator=g), torch.randn(N, 3, H, W, generator=g)
    ones = torch.ones(N, 1, H, W)
    m = torch.tensor(obox.generate_params(N, (H, W), 0.5, invert=True, rng=np.random.RandomState(0)).astype(np.float32))
    if 'v3plus' in workload.get('arch', ''):
        from oracle import deeplab3plus as o3, step_v3plus as sv
        N = 2                                   # batch statistics in the head need more than one sample
        x, y, ux0, ux1, ones, m = (torch.cat([t[:1], t[:1].flip(3)], 0) for t in (x, y, ux0, ux1, ones, m))
        S = sv.StepStateV3Plus(o3.closed_form_state(C), C, lr=3e-5)
        run = lambda: sv.train_iteration(S, x, y, ux0, ux1, ones, ones, m)
        what = 'oracle/step_v3plus.py'
    else:
        S = ostep.StepState(odl.closed_form_state(C), C, opt='adam', lr=3e-5)
        run = lambda: ostep.train_iteration(S, x, y, ux0, ux1, ones, ones, m)
        what = 'oracle/step.py'
    min_iters = 3
    t_w = time.time()
    run()                                                             # warm-up
    if time.time() - t_w > 2.0 * seconds_budget / 3.0:                # a slow / busy host: do not run for minutes
        min_iters = 1
    times = []
    t_start = time.time()
    while len(times) < min_iters or (time.time() - t_start < seconds_budget * 0.5 and len(times) < 8):
        t0 = time.time()
        run()
        times.append(time.time() - t0)
    t = sum(times) / len(times)
    return dict(value=N / t, unit='images/sec', cores=cores, kind='port',
                sample=what + ' (PyTorch-CPU fp32 restatement of the reference step), batch {} of {}x{} '
                       '(GPU run: batch {}), {} timed iterations after 1 warm-up, {:.2f} s/iter (min {:.2f}, max {:.2f})'.format(
                           N, H, W, workload['batch'], len(times), t, min(times), max(times)))


def measure_traffic(key):
    """-> {'traffic': HBM bytes per convolution launch or None, 'traffic_source': how}. Runs `bench.py --workload <key>
    --steps 2 --timed_only --no_overlap` twice under `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE, then
    WRITE_SIZE: they do not fit one pass; kernel trace only, serial streams so that a dispatch's counters are its own) and
    averages the counters over the conv_igemm launches. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts a
    128-byte read request as 64 bytes -> doubled. Any failure (no rocprofv3, counters unavailable, timeout) gives None with
    the reason -- never a number from a file."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rocprof):
        return {'traffic': None, 'traffic_source': 'rocprofv3 not found'}
    kb = {}
    launches = 0
    per_kernel = {}            # rocprofv3 kernel name -> {counter: [sum of KB, launches]}
    sub_by_kernel = None       # algorithmic bytes per kernel of the profiled (serial-stream) run itself, from its own JSON line
    tmp = tempfile.mkdtemp(prefix='cms_pmc_', dir='/tmp')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            outdir = os.path.join(tmp, counter)
            cmd = [rocprof, '--kernel-trace', '--pmc', counter, '-d', outdir, '-o', 'p', '--output-format', 'csv', '--',
                   sys.executable, os.path.abspath(__file__), '--workload', key, '--steps', '2', '--warmup', '1',
                   '--no_cpu_baseline', '--no_overlap', '--no_roofline_events', '--timed_only', '--traffic', 'omit',
                   '--detail_stdout', '--detail_path', os.path.join(outdir + '_detail.json')]
            env = dict(os.environ, TMPDIR='/tmp')
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420)
            vals = []
            for f in glob.glob(os.path.join(outdir, '**', '*counter_collection.csv'), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get('Counter_Name') != counter:
                        continue
                    kn = row.get('Kernel_Name', '')
                    if 'conv_igemm' in kn or 'conv8_kernel' in kn:
                        vals.append(float(row['Counter_Value']))
                    if 'cms::' in kn:
                        acc = per_kernel.setdefault(kn, {}).setdefault(counter, [0.0, 0])
                        acc[0] += float(row['Counter_Value']); acc[1] += 1
            if sub_by_kernel is None:
                for line in r.stdout.decode(errors='replace').splitlines():
                    if line.startswith('{"bench_detail"'):
                        try:
                            sub_by_kernel = json.loads(line)['bench_detail']['roofline'].get('by_kernel')
                        except Exception:              # noqa: BLE001
                            sub_by_kernel = None
            if r.returncode != 0 or not vals:
                return {'traffic': None, 'traffic_source': 'rocprofv3 --pmc {} pass gave no conv_igemm rows (rc {})'.format(
                    counter, r.returncode)}
            kb[counter] = sum(vals) / len(vals)
            launches = len(vals)
    except Exception as e:                                   # noqa: BLE001 (a profiler hiccup must not lose the bench line)
        return {'traffic': None, 'traffic_source': 'traffic measurement failed: {}'.format(type(e).__name__)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    traffic = (2.0 * kb['FETCH_SIZE'] + kb['WRITE_SIZE']) * 1024.0
    # per kernel: counter bytes per launch beside the algorithmic bytes per launch of the SAME (serial-stream) run -> which
    # kernel over-fetches (ratio > 1: re-reads of tiles from HBM, partial-line writes; < 1: operands served by the L2 / MALL)
    by_kernel = {}
    for kn, cs in per_kernel.items():
        if 'FETCH_SIZE' not in cs or 'WRITE_SIZE' not in cs:
            continue
        pmc = (2.0 * cs['FETCH_SIZE'][0] / cs['FETCH_SIZE'][1] + cs['WRITE_SIZE'][0] / cs['WRITE_SIZE'][1]) * 1024.0
        entry = {'pmc_bytes_per_launch': pmc, 'launches': cs['FETCH_SIZE'][1],
                 'fetch_bytes_per_launch': 2.0 * cs['FETCH_SIZE'][0] / cs['FETCH_SIZE'][1] * 1024.0,
                 'write_bytes_per_launch': cs['WRITE_SIZE'][0] / cs['WRITE_SIZE'][1] * 1024.0}
        for route, info in (sub_by_kernel or {}).items():
            if route in kn:
                entry['algorithmic_bytes_per_launch'] = info['algorithmic_bytes_per_launch']
                entry['ratio'] = pmc / max(info['algorithmic_bytes_per_launch'], 1.0)
        by_kernel[kn[:100]] = entry
    return {'traffic': traffic, 'traffic_launches': launches, 'traffic_by_kernel': by_kernel,
            'traffic_fetch_kb_per_launch': kb['FETCH_SIZE'], 'traffic_write_kb_per_launch': kb['WRITE_SIZE'],
            'traffic_source': 'measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes of '
                              'bench.py --workload {} --steps 2 --no_overlap), averaged over the conv_igemm_* / conv8_kernel launches; gfx950: '
                              'FETCH_SIZE doubled (128-byte requests counted at 64)'.format(key)}


def self_launch(args):
    """`--gpus N` without a launcher: become `python -m torch.distributed.run ... bench.py <same flags>`."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL needs on this driver
    sys.stdout.flush()
    os.execvp(cmd[0], cmd)


def dry_launch(args, world, rank):
    """Launch-path check that needs no GPU: every rank joins a gloo group, an all-reduce counts the ranks, rank 0
    prints the JSON skeleton with the world size it saw (tests/test_dist_cpu.py drives `--gpus 2 --dry_launch`)."""
    import torch
    import torch.distributed as dist
    seen = 1
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.barrier()
    if rank == 0:
        print(json.dumps({'metric': 'train images/sec (student+teacher step)', 'value': None, 'unit': 'images/sec',
                          'n_gpus': world, 'world_size_seen': seen, 'gpus_requested': args.gpus, 'dry_launch': True,
                          'steps': args.steps, 'warmup': args.warmup}))
    if world > 1:
        dist.destroy_process_group()


def run_workload(key, args, world, rank, dev):
    """Builds the networks of workload `key`, times K steps, returns the result object (meaningful on rank 0)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from cutmix_semisup_seg_amd import ops, optim as fo
    from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig, UnsupBatch
    from architectures import network_architectures
    import mask_gen
    import optim_weight_ema

    wl = WORKLOADS[key]
    B, H, W, C = wl['batch'], wl['H'], wl['W'], wl['classes']
    if os.environ.get('CMS_BENCH_BATCH'):       # EXPERIMENT (tile-quantisation studies): another batch size -- not the headline config
        B = int(os.environ['CMS_BENCH_BATCH'])
    if os.environ.get('CMS_BENCH_HW_' + key.upper()):       # EXPERIMENT ONLY (tile-fit studies): another crop size, e.g. 313x313
        H, W = (int(v) for v in os.environ['CMS_BENCH_HW_' + key.upper()].split('x'))
        wl = dict(wl, H=H, W=W, name=wl['name'] + ' [EXPERIMENT: crop {}x{}]'.format(H, W))
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    esz = 2.0 if dtype == torch.bfloat16 else 4.0
    roofline_kernel = args.roofline_kernel or 'conv'       # the dominant kernel of every workload (incl. DeepLab v3+)
    sample_every = max(1, args.roofline_sample)

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
    stu.train(); tea.train()
    if not args.no_freeze_bn:
        stu.freeze_batchnorm(); tea.freeze_batchnorm()      # --freeze_bn (run_*_experiments.sh)
    cfg = StepConfig(mask_mode='mix', cons_loss_fn='var', cons_weight=1.0, conf_thresh=0.97, conf_per_pixel=False,
                     fuse_batches=not args.no_fuse_batches, compute_dtype=dtype,
                     overlap_teacher=not args.no_overlap, allreduce_dtype=args.allreduce_dtype,
                     deterministic=args.deterministic, early_optimizer=args.early_optimizer)
    step = CutMixMeanTeacherStep(stu, tea, opt, ema, cfg)
    step.time_buckets = world > 1            # (bytes, issue-to-wait) of every gradient bucket of the LAST timed step
    has_ex = hasattr(stu, 'hip_executor')
    if args.no_overlap and has_ex:
        stu.hip_executor().overlap_wgrad = False
    if args.wgrad_streams and has_ex:
        stu.hip_executor().wgrad_streams = args.wgrad_streams
    if args.conv_tile and has_ex:
        stu.hip_executor().conv_tile = tea.hip_executor().conv_tile = args.conv_tile
    if args.tile_rule and has_ex:
        rules = {int(a): int(b) for a, b in (kv.split(':') for kv in args.tile_rule.split(','))}
        stu.hip_executor().tile_rules = tea.hip_executor().tile_rules = rules
    if os.environ.get('CMS_TEACHER_TILE_RULE') and has_ex:
        # EXPERIMENT (round 6): the TEACHER's K-deep convolutions on the 128 x 128 family (e.g. "256-128+512-128") while the student's
        # stay on the eight-phase kernel -- two eight-phase launches of 132 tiles are 264 whole-CU workgroups for 256 CUs
        tea.hip_executor().tile_rules = {int(a): int(b) for a, b in (kv.split('-') for kv in os.environ['CMS_TEACHER_TILE_RULE'].split('+'))}

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

    # ---- roofline instrumentation: HIP events on the launch stream around the launches of the measured kernels
    timing_on = [False]
    bracket_on = [True]            # event brackets armed for the current step (the work counters run on every timed step)
    conv = dict(pairs=[], work=[], flops=0.0, bytes=0.0, launches=0)       # conv_igemm_kernel (MFMA)
    hbm = dict(pairs=[], bytes=0.0, moved=0.0, parts={})                   # the HBM-bound group
    other = dict(pairs=[], work=[])                                        # --roofline_kernel adam_ema / consistency
    saved_fns = {}

    def ev_pair():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def hbm_timed(part, fn, nbytes, moved):
        def wrapper(*a, **k):
            if not timing_on[0] or args.no_roofline_events or not bracket_on[0]:
                return fn(*a, **k)
            e0, e1 = ev_pair()
            e0.record()
            r = fn(*a, **k)
            e1.record()
            nb, mv = nbytes(*a, **k), moved(*a, **k)
            hbm['pairs'].append((part, e0, e1))
            hbm['bytes'] += nb
            hbm['moved'] += mv
            p = hbm['parts'].setdefault(part, [0.0, 0.0, 0])
            p[0] += nb; p[1] += mv; p[2] += 1
            return r
        return wrapper

    P_ = lambda n: float(n) * H * W                           # full-resolution pixels of n images

    def lowres(t):
        return float(t.shape[0]) * t.shape[2] * t.shape[3]    # low-resolution cells of an (N,C,h,w) tensor

    # SURVEY.md 8(d), compulsory bytes of the reference-equivalent (materialised) formulation / bytes the fused kernels move
    saved_fns['cutmix_paste'] = ops.cutmix_paste
    ops.cutmix_paste = hbm_timed('cutmix_paste', saved_fns['cutmix_paste'],
                                 lambda x0, x1, **k: 9.0 * P_(x1.shape[0]) * esz,      # 2 image reads + 1 write (um all-ones)
                                 lambda x0, x1, **k: 9.0 * P_(x1.shape[0]) * esz)
    saved_fns['consistency_forward'] = ops.consistency_forward
    ops.consistency_forward = hbm_timed('consistency_fwd', saved_fns['consistency_forward'],
                                        lambda cfg_, ls, *a, **k: (2.0 * C + 2.0) * P_(ls.shape[0]) * 4.0,
                                        lambda cfg_, ls, *a, **k: 3.0 * C * lowres(ls) * 4.0)
    saved_fns['consistency_backward'] = ops.consistency_backward
    def _frac(ctx, k):             # a launch over a run of the samples (step.py issues the backward as two halves on two streams)
        sm = k.get('samples')
        return 1.0 if sm is None else float(sm[1] - sm[0]) / float(ctx[1][0].shape[0])
    ops.consistency_backward = hbm_timed('consistency_bwd', saved_fns['consistency_backward'],
                                         lambda ctx, sc, *a, **k: (3.0 * C + 2.0) * P_(ctx[1][0].shape[0]) * 4.0 * _frac(ctx, k),
                                         lambda ctx, sc, *a, **k: 4.0 * C * lowres(ctx[1][0]) * 4.0 * _frac(ctx, k))
    saved_fns['ce_forward'] = ops.ce_forward
    ops.ce_forward = hbm_timed('ce_fwd', saved_fns['ce_forward'],
                               lambda lg, lab, *a, **k: C * P_(lg.shape[0]) * 4.0 + P_(lg.shape[0]) * 1.0,
                               lambda lg, lab, *a, **k: C * lowres(lg) * 4.0 + P_(lg.shape[0]) * 1.0)
    saved_fns['ce_backward'] = ops.ce_backward
    ops.ce_backward = hbm_timed('ce_bwd', saved_fns['ce_backward'],
                                lambda ctx, sc, *a, **k: 2.0 * C * P_(ctx[1][0].shape[0]) * 4.0 + P_(ctx[1][0].shape[0]),
                                lambda ctx, sc, *a, **k: 2.0 * C * lowres(ctx[1][0]) * 4.0 + P_(ctx[1][0].shape[0]))

    # (round 6) each loss as ONE launch: the same compulsory bytes as its forward + backward pair ((5C + 4) P s for the
    # consistency loss, 3 C P s + 2 P label bytes for the cross entropy), the rectangles staged once instead of twice
    saved_fns['consistency_fused'] = ops.consistency_fused
    ops.consistency_fused = hbm_timed('consistency_fwd_bwd', saved_fns['consistency_fused'],
                                      lambda cfg_, ls, *a, **k: (5.0 * C + 4.0) * P_(ls.shape[0]) * 4.0,
                                      lambda cfg_, ls, *a, **k: 4.0 * C * lowres(ls) * 4.0)
    saved_fns['ce_fused'] = ops.ce_fused
    ops.ce_fused = hbm_timed('ce_fwd_bwd', saved_fns['ce_fused'],
                             lambda lg, lab, *a, **k: 3.0 * C * P_(lg.shape[0]) * 4.0 + 2.0 * P_(lg.shape[0]),
                             lambda lg, lab, *a, **k: 2.0 * C * lowres(lg) * 4.0 + P_(lg.shape[0]))

    orig_conv = saved_fns['conv_igemm'] = ops.conv_igemm
    orig_wgrad = saved_fns['conv_wgrad'] = ops.conv_wgrad

    def conv_flops(x, w_packed, k):
        ohw = k.get('out_hw')
        npix = x.shape[0] * (ohw[0] * ohw[1] if ohw is not None else x.shape[1] * x.shape[2])
        return 2.0 * npix * w_packed.shape[1] * w_packed.shape[2] * w_packed.shape[0]

    def timed_conv(x, w_packed, taps, *a, **k):
        if not timing_on[0]:
            return orig_conv(x, w_packed, taps, *a, **k)
        fl = conv_flops(x, w_packed, k)
        conv['flops'] += fl
        npix_out = fl / (2.0 * w_packed.shape[1] * w_packed.shape[2] * w_packed.shape[0])
        head = k.get('out_f32_nchw') is not None
        nb = esz * (x.numel() + w_packed.numel()) + npix_out * (
            (4.0 * k.get('cout_real', w_packed.shape[1]) if head else esz * w_packed.shape[1])
            + (esz * w_packed.shape[1] if k.get('res') is not None else 0.0)
            + (esz * w_packed.shape[1] if k.get('mask_src') is not None else 0.0))
        conv['bytes'] += nb
        conv['launches'] += 1
        if args.no_roofline_events or not bracket_on[0]:
            return orig_conv(x, w_packed, taps, *a, **k)
        if head:            # the ASPP head (2048 -> C, dilations 6 + 12): HBM-bound group, every launch timed
            e0, e1 = ev_pair()
            e0.record()
            r = orig_conv(x, w_packed, taps, *a, **k)
            e1.record()
            hbm['pairs'].append(('aspp_head_fwd', e0, e1))
            hbm['bytes'] += nb
            hbm['moved'] += nb
            p = hbm['parts'].setdefault('aspp_head_fwd', [0.0, 0.0, 0])
            p[0] += nb; p[1] += nb; p[2] += 1
            return r
        if roofline_kernel != 'conv' or conv['launches'] % sample_every:
            return orig_conv(x, w_packed, taps, *a, **k)
        e0, e1 = ev_pair()
        e0.record()
        r = orig_conv(x, w_packed, taps, *a, **k)
        e1.record()
        conv['work'].append(fl)
        conv['pairs'].append((e0, e1))
        return r

    def counted_wgrad(du, x, taps, dw, *a, **k):
        if timing_on[0]:
            conv['flops'] += 2.0 * du.shape[0] * du.shape[1] * du.shape[2] * du.shape[3] * x.shape[3] * len(taps)
        return orig_wgrad(du, x, taps, dw, *a, **k)
    ops.conv_wgrad = counted_wgrad
    ops.conv_igemm = timed_conv

    kname = ('cms_conv_igemm launches = conv8_kernel (eight-phase 256x256 tile, the K-deep layers) + conv_igemm_*kernel (128x128 '
             'tile): MFMA implicit-GEMM convolution, forward + data-gradient launches of the backbone')
    roof = dict(bound='mfma', peak=MFMA_PEAK_TFLOPS if dtype == torch.bfloat16 else MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s')
    from cutmix_semisup_seg_amd import _lib
    orig_adam = _lib.fn['cms_adam_ema_step']
    if roofline_kernel == 'adam_ema':
        def timed_launch(desc, stream):
            if not timing_on[0]:
                return orig_adam(desc, stream)
            e0, e1 = ev_pair()
            e0.record()
            rc = orig_adam(desc, stream)
            e1.record()
            other['pairs'].append((e0, e1))
            other['work'].append(opt.arena.total * 40.0)   # 5 fp32 reads + 4 fp32 writes + 2 bf16 copies
            return rc
        _lib.fn['cms_adam_ema_step'] = timed_launch
        kname = 'optim_ema_kernel<ADAM> (fused Adam + teacher EMA over the 44.2M-element arena)'
        roof = dict(bound='hbm', peak=HBM_PEAK_GBS, unit='GB/s')

    # The DeepLab v2 body runs as recorded launch programs (csrc/program.hip): its convolution launches never pass
    # through ops.conv_igemm at run time. The program runner brackets every k-th of them (and every ASPP head launch)
    # with HIP events on the launch stream itself; the Python wrappers above only see eagerly issued launches (v3+).
    def executors():
        if not has_ex:
            return []
        from cutmix_semisup_seg_amd.backbone_hip import executors_of
        return [e for net in (stu, tea) for e in executors_of(net) if e.use_programs]

    def arm_timing(k):
        for e in executors():
            for pr in e.programs():
                pr.set_timing(0 if args.no_roofline_events else k)

    def read_timing():
        tot = dict(ms=0.0, flops=0.0, launches=0, head_ms=0.0, head_launches=0)
        for e in executors():
            for pr in e.programs():
                t = pr.read_timing()
                for kk in tot:
                    tot[kk] += t[kk]
        return tot

    def issued():
        tot = dict(flops=0.0, conv_launches=0, conv_bytes=0.0, head_launches=0, head_bytes=0.0, head_bytes_alg=0.0, floor_s=0.0)
        for e in executors():
            for kk in tot:
                tot[kk] += e.issued.get(kk, 0.0)
        return tot

    def by_route():
        tot = {}
        for e in executors():
            for k, v in e.issued.get('by_route', {}).items():
                r = tot.setdefault(k, [0, 0.0, 0.0])
                r[0] += v[0]; r[1] += v[1]; r[2] += v[2]
        return tot

    def one_step(i):
        b = pool[i % len(pool)]
        ranges = ops.ranges_to_device(boxgen.generate_ranges(B, (H, W), rng=mask_rng), dev)
        ub = UnsupBatch(b['x0'], ranges, x1_tea=b['x1'], x0_stu=b['x0s'], x1_stu=b['x1s'])
        return step(b['x'], b['y'], [ub])

    try:
        for i in range(args.warmup):
            one_step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        timing_on[0] = True
        read_timing()                                # drop whatever the warm-up left
        arm_timing(sample_every if roofline_kernel == 'conv' else 0)
        issued0 = issued()
        route0 = by_route()
        # Event brackets cost throughput (a bracketed launch cannot overlap its neighbours on the queue: 626-629 img/s with every 5th
        # convolution launch of EVERY step bracketed against 636 with none, profiles/r05h_*, r05i_*): they are armed on every
        # `--roofline_steps_every`-th timed step only (default 4) -- still a few hundred timed launches inside the timed region
        every_s = max(1, args.roofline_steps_every)
        t0 = time.perf_counter()
        for i in range(args.steps):
            armed = (i % every_s) == 0
            if every_s > 1:
                bracket_on[0] = armed
                arm_timing((sample_every if roofline_kernel == 'conv' else 0) if armed else 0)
            res = one_step(i)
        bracket_on[0] = True
        t_enqueue = time.perf_counter() - t0         # host time to enqueue the K steps (== elapsed when launch-bound)
        torch.cuda.synchronize()
        t_local = time.perf_counter() - t0           # this rank's own K steps (before it waits for the slowest rank)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timing_on[0] = False
        arm_timing(0)
        prog_t = read_timing()
        issued1 = issued()
        prog_i = {k: issued1[k] - issued0[k] for k in issued1}
        route1 = by_route()
        routes = {k: [v[j] - route0.get(k, [0, 0.0, 0.0])[j] for j in range(3)] for k, v in route1.items()}
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax)
        # per-rank view (the first SCALE run should be diagnosable from one line): every rank's own time for the K steps and
        # the slowest wait of every gradient bucket over the ranks
        rank_view = {'local_ms_per_step': [1e3 * t_local / args.steps]}
        buckets = step.bucket_timing()
        if world > 1:
            tl = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(tl, torch.tensor([1e3 * t_local / args.steps], dtype=torch.float64, device=dev))
            rank_view['local_ms_per_step'] = [float(v) for v in tl]
            if buckets:
                waits = torch.tensor([float(b['issue_to_wait_ms']) for b in buckets], dtype=torch.float64, device=dev)
                wmax = waits.clone()
                dist.all_reduce(wmax, op=dist.ReduceOp.MAX)
                rank_view['bucket_wait_ms_max_over_ranks'] = [float(v) for v in wmax]
        rank_view['min'] = min(rank_view['local_ms_per_step'])
        rank_view['max'] = max(rank_view['local_ms_per_step'])

        last = {k: (None if v is None else float(v)) for k, v in res.items()}
        if not np.isfinite(last['sup_loss']):
            raise SystemExit('non-finite loss in the bench loop: {}'.format(last))

        # Host cost of enqueueing ONE step with an empty launch queue (in the timed loop the enqueue call blocks on the
        # queue whenever the GPU is the limit, so `host_enqueue_ms_per_step` ~ `ms_per_step` says nothing about the host)
        host_ms = []
        for i in range(0 if args.timed_only else 4):
            torch.cuda.synchronize()
            th = time.perf_counter()
            one_step(i)
            host_ms.append(1e3 * (time.perf_counter() - th))
        torch.cuda.synchronize()
        host_unblocked = float(np.median(host_ms)) if host_ms else None
        if args.copy_profile and rank == 0:          # which python lines issue device-to-device copies / fills (stderr)
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
                for i in range(2):
                    one_step(i)
                torch.cuda.synchronize()
            rows = [e for e in prof.key_averages(group_by_stack_n=8)
                    if ('copy_' in e.key or 'clone' in e.key or 'Memcpy' in e.key or 'fill_' in e.key or 'zero_' in e.key)]
            rows.sort(key=lambda e: -e.count)
            for e in rows[:40]:
                sys.stderr.write('{:6d} x {:28s} cuda {:9.1f} us | {}\n'.format(
                    e.count, e.key[:28], e.device_time_total if hasattr(e, 'device_time_total') else 0.0,
                    ' <- '.join(str(f).split('/')[-1] for f in e.stack[:5])))
        if args.host_profile and rank == 0:          # where the host time of a step goes (stderr; not part of the line)
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            for i in range(3):
                one_step(i)
                torch.cuda.synchronize()
            pr.disable()
            st_ = pstats.Stats(pr, stream=sys.stderr)
            st_.sort_stats('cumulative').print_stats(45)
            st_.sort_stats('tottime').print_stats(30)

        # Outside the timed region: the same kernel with the GPU to itself (single stream, every launch bracketed). In
        # the timed region the teacher pass and the weight gradients run concurrently on other streams, so a launch's
        # event-to-event time there includes the share of the machine the co-running kernels took.
        isolated = None
        timed = dict(pairs=list(conv['pairs']), work=list(conv['work']), flops=conv['flops'], bytes=conv['bytes'],
                     launches=conv['launches'])
        hbm_timed_pairs = list(hbm['pairs'])
        hbm_tot = (hbm['bytes'], hbm['moved'], {k: list(v) for k, v in hbm['parts'].items()})
        if world == 1 and roofline_kernel == 'conv' and not args.no_overlap and not args.no_roofline_events and has_ex \
                and 'arch' not in wl and not args.timed_only:
            del conv['pairs'][:], conv['work'][:]
            cfg.overlap_teacher = False
            stu.hip_executor().overlap_wgrad = False
            sample_every = 1
            one_step(0)                              # records the single-stream programs
            torch.cuda.synchronize()
            read_timing()
            arm_timing(1)
            timing_on[0] = True
            for i in range(3):
                one_step(i)
            torch.cuda.synchronize()
            timing_on[0] = False
            arm_timing(0)
            it = read_timing()
            ms_iso = [a.elapsed_time(b) for a, b in conv['pairs']]
            tot_ms, tot_fl, tot_n = sum(ms_iso) + it['ms'], sum(conv['work']) + it['flops'], len(ms_iso) + it['launches']
            ach = tot_fl / (tot_ms * 1e-3) / 1e12
            isolated = {'achieved': ach, 'frac': ach / roof['peak'], 'avg_launch_ms': tot_ms / max(tot_n, 1),
                        'launches_timed': tot_n, 'note': 'same kernel, single stream, 3 extra steps after the '
                        'timed region (not part of `value`)'}
    finally:
        for name, f in saved_fns.items():
            setattr(ops, name, f)
        _lib.fn['cms_adam_ema_step'] = orig_adam

    out = None
    if rank == 0:
        if roofline_kernel == 'conv':
            pairs, work = timed['pairs'], timed['work']
        else:
            pairs, work = other['pairs'], other['work']
        ms_all = [a.elapsed_time(b) for a, b in pairs]
        sum_ms, sum_work, n_timed = sum(ms_all), sum(work), len(ms_all)
        if roofline_kernel == 'conv':               # + the launches the program runner bracketed
            sum_ms, sum_work, n_timed = sum_ms + prog_t['ms'], sum_work + prog_t['flops'], n_timed + prog_t['launches']
            timed['flops'] += prog_i['flops']
            timed['bytes'] += prog_i['conv_bytes']
            timed['launches'] += prog_i['conv_launches']
        ms_kernel = sum_ms / n_timed if n_timed else float('nan')
        per_launch = sum_work / n_timed if n_timed else float('nan')
        rate = (sum_work / (sum_ms * 1e-3)) if sum_ms > 0 else float('nan')
        achieved = rate / (1e12 if roof['bound'] == 'mfma' else 1e9)
        out = {
            'workload': key,
            'value': args.steps * B * world / elapsed,
            'unit': 'images/sec',
            'ms_per_step': 1e3 * elapsed / args.steps,
            'config': {'workload': wl['name'], 'per_gpu_batch': B, 'global_batch': B * world, 'crop': [H, W],
                       'classes': C, 'parallelism': 'dp{}'.format(world), 'image_forwards_per_sec':
                           4 * args.steps * B * world / elapsed,
                       'fuse_batches': not args.no_fuse_batches, 'stream_overlap': not args.no_overlap,
                       'host_enqueue_ms_per_step': 1e3 * t_enqueue / args.steps,
                       'host_enqueue_ms_per_step_empty_queue': host_unblocked,
                       'last_losses': last,
                       'allreduce': {'dtype': args.allreduce_dtype, 'buckets_last_step': buckets},
                       'ranks': rank_view,
                       'deterministic_wgrad': bool(args.deterministic),
                       # (device, ms of the spin kernel alone, clash matrix of {default stream, six candidates}: 1 = the pair shares a
                       # hardware queue, candidates chosen as side streams): ops._probe_side_streams
                       'stream_probe': [list(p_) for p_ in ops._STREAM_PROBE_LOG],
                       'parity_config': ('this line is the bf16-STORAGE engine: held per layer (teacher-forced) to the bf16-storage '
                                         'oracle at 1.5e-4 forward / 2e-3 backward / 3e-4 weight gradients; the 1e-4 bar on whole-'
                                         'iteration losses / IoU is held by the fp32 hand-written engine (--dtype fp32), which is '
                                         'not the timed configuration (DESIGN.md 2.1)') if args.dtype == 'bf16' else
                                        'fp32 hand-written engine: whole-iteration losses / IoU within 1e-4 of the oracle'},
            'roofline': {'bound': roof['bound'], 'kernel': kname, 'achieved': achieved, 'peak': roof['peak'],
                         'unit': roof['unit'], 'frac': achieved / roof['peak'], 'traffic': None,
                         'avg_launch_ms': ms_kernel,
                         'algorithmic_{}_per_launch'.format('flops' if roof['bound'] == 'mfma' else 'bytes'): per_launch,
                         'launches_timed': n_timed,
                         'sampling': ('every launch' if args.roofline_sample <= 1 else
                                      'every {}th launch'.format(args.roofline_sample)) +
                                     ('' if args.roofline_steps_every <= 1 else
                                      ' of every {}th timed step'.format(args.roofline_steps_every))},
        }
        if routes:
            # algorithmic work per KERNEL of the recorded launches (the library's own routing, cms_conv_igemm_route /
            # cms_conv_wgrad_uses_wgrad8): what `traffic_by_kernel` puts beside the PMC counters
            out['roofline']['by_kernel'] = {k: {'launches_per_step': v[0] / args.steps,
                                                'algorithmic_bytes_per_launch': v[1] / max(v[0], 1),
                                                'flops_per_launch': v[2] / max(v[0], 1)} for k, v in sorted(routes.items()) if v[0] > 0}
        if roofline_kernel == 'conv' and timed['launches']:
            out['roofline']['algorithmic_bytes_per_launch'] = timed['bytes'] / timed['launches']
            if key == 'pascal' and getattr(args, 'traffic', 'omit') == 'measure' and world == 1:
                # HBM bytes per launch of THIS build's kernels, measured now: two more runs of this script (2 steps each)
                # under rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM / PMC slots)
                out['roofline'].update(measure_traffic(key))
            else:
                out['roofline']['traffic_source'] = 'not measured in this run (--traffic measure, single GPU, workload pascal)'
        if isolated is not None:
            out['roofline']['isolated'] = isolated
        if prog_i.get('floor_s', 0.0) > 0:
            # the MIXED roofline beside the MFMA-only fraction. Per recorded convolution / weight-gradient launch
            # max(algorithmic bytes / 8 TB/s, FLOPs / 2.5 PFLOP/s), summed over the step: the time the step's convolution work
            # would take if every launch ran alone at whichever roof binds it
            out['roofline']['mixed'] = {
                'floor_ms_per_step': prog_i['floor_s'] / args.steps * 1e3,
                'step_ms': elapsed / args.steps * 1e3,
                'frac': prog_i['floor_s'] / elapsed,
                'basis': 'sum over the recorded convolution, ASPP-head and weight-gradient launches of max(bytes / 8 TB/s, '
                         'FLOPs / 2.5 PFLOP/s), over the step time'}
        if timed['flops'] > 0:
            # all MFMA work of the step (forward, data-gradient AND weight-gradient convolutions) over the step time
            out['roofline']['step_mfma'] = {
                'tflop_per_step': timed['flops'] / args.steps / 1e12,
                'achieved': timed['flops'] / elapsed / 1e12, 'frac': timed['flops'] / elapsed / 1e12 / roof['peak']}
        if hbm_timed_pairs or prog_t['head_launches']:
            ms = {}
            for part, e0, e1 in hbm_timed_pairs:
                ms[part] = ms.get(part, 0.0) + e0.elapsed_time(e1)
            nb, mv, parts = hbm_tot
            if prog_t['head_launches']:              # ASPP head launches issued by the program runner
                ms['aspp_head_fwd'] = ms.get('aspp_head_fwd', 0.0) + prog_t['head_ms']
                # the head's ALGORITHMIC bytes are the 2048-channel input once + weights + logits; the fp32 Z planes
                # the single-pass formulation writes and re-reads are counted as MOVED bytes only
                share = prog_t['head_launches'] / max(prog_i['head_launches'], 1)
                hb, hm = prog_i['head_bytes_alg'] * share, prog_i['head_bytes'] * share
                nb, mv = nb + hb, mv + hm
                pp = parts.setdefault('aspp_head_fwd', [0.0, 0.0, 0])
                pp[0] += hb; pp[1] += hm; pp[2] += prog_t['head_launches']
            tot_ms = sum(ms.values())
            n_br = max(1, (args.steps + max(1, args.roofline_steps_every) - 1) // max(1, args.roofline_steps_every))   # bracketed steps
            out['roofline_hbm'] = {
                'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'group': 'cutmix paste + masked consistency fwd+bwd + cross entropy fwd+bwd + ASPP head convolution',
                'achieved': nb / (tot_ms * 1e-3) / 1e9, 'frac': nb / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'basis': 'SURVEY.md 8(d): compulsory bytes of the reference-equivalent formulation (full-resolution '
                         'logits materialised); the fused kernels evaluate the bilinear upsample in-kernel and move '
                         '`moved_bytes_per_step` instead',
                'bytes_per_step': nb / n_br, 'moved_bytes_per_step': mv / n_br,
                'achieved_on_moved_bytes': mv / (tot_ms * 1e-3) / 1e9,
                'ms_per_step': tot_ms / n_br, 'steps_bracketed': n_br,
                'parts': {k: {'ms_per_step': ms[k] / n_br, 'GBps': parts[k][0] / (ms[k] * 1e-3) / 1e9,
                              'GBps_moved': parts[k][1] / (ms[k] * 1e-3) / 1e9,
                              'launches_per_step': parts[k][2] / n_br} for k in ms},
                'traffic': None}
        # scalar copies of the nested objects (the driver's record keeps scalar keys of `roofline` / `config` only)
        rf = out['roofline']
        for name in ('mixed', 'step_mfma', 'isolated'):
            if isinstance(rf.get(name), dict):
                rf[name + '_frac'] = rf[name]['frac']
        if 'roofline_hbm' in out:
            rf['hbm_group_frac'] = out['roofline_hbm']['frac']
            rf['hbm_group_ms_per_step'] = out['roofline_hbm']['ms_per_step']
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(wl, seconds_budget=30.0 if key == 'pascal' else 20.0)
    del step, opt, ema, stu, tea, pool
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', choices=sorted(WORKLOADS) + ['both'], default='both',
                    help='both = pascal (321x321, the headline value) then cityscapes (512x1024), one JSON line')
    ap.add_argument('--dtype', choices=['bf16', 'fp32'], default='bf16')
    ap.add_argument('--roofline_kernel', choices=['conv', 'adam_ema'], default=None,
                    help='default: conv (the MFMA convolution) for the DeepLab v2 workloads, adam_ema otherwise')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--traffic', choices=['measure', 'omit'], default='measure',
                    help='roofline.traffic: measure it now with two rocprofv3 --pmc passes of this script (single GPU, '
                         'workload pascal; ~1.5 min), or leave it null')
    ap.add_argument('--no_fuse_batches', action='store_true')
    ap.add_argument('--no_freeze_bn', action='store_true',
                    help='BatchNorm on batch statistics (the reference CLI default; its experiment scripts pass --freeze_bn): '
                         'the passes run layer by layer through the conv engine instead of the static executor')
    ap.add_argument('--allreduce_dtype', choices=['fp32', 'bf16'], default='fp32',
                    help='data-parallel gradient exchange: the fp32 arena (177 MB per step) or a bf16 staging copy (86 MB)')
    ap.add_argument('--deterministic', action='store_true',
                    help='run-to-run deterministic weight gradients (slab + ordered reduce instead of fp32 atomics)')
    ap.add_argument('--conv_tile', type=int, default=0, help='experiment: force a conv tile code (256, 1128, 128)')
    ap.add_argument('--tile_rule', default='', help='experiment: cout:tile[,cout:tile...] per-layer tile codes')
    ap.add_argument('--no_overlap', action='store_true', help='single stream: no teacher / weight-gradient overlap')
    ap.add_argument('--wgrad_streams', type=int, default=0, help='experiment: streams the weight gradients are spread over')
    ap.add_argument('--no_roofline_events', action='store_true', help='skip the per-launch event brackets')
    ap.add_argument('--timed_only', action='store_true',
                    help='skip the extra single-stream and empty-queue steps after the timed region (so that a rocprofv3 '
                         '--stats run of this command averages in-step launches only)')
    ap.add_argument('--copy_profile', action='store_true', help='torch.profiler: python sources of the device copies / fills of two steps (stderr)')
    ap.add_argument('--host_profile', action='store_true', help='cProfile of three steps after the timed region (stderr)')
    ap.add_argument('--roofline_steps_every', type=int, default=4,
                    help='arm the event brackets (convolution launches, HBM group) on every k-th timed step only')
    ap.add_argument('--roofline_sample', type=int, default=5,
                    help='bracket every k-th launch of the conv kernel with events (1 = all; the brackets cost ~3 %% '
                         'of the step when every launch carries them)')
    ap.add_argument('--early_optimizer', action='store_true',
                    help='A/B: optimizer + EMA launches per finished gradient slice during the backward pass (third stream)')
    ap.add_argument('--no_also', action='store_true',
                    help='default run (--workload both, one GPU): skip the two extra short runs reported under "also" '
                         '(DeepLab v3+ at configs[3]; configs[1] without --freeze_bn)')
    ap.add_argument('--detail_path', default='', help='where the complete result object goes (default: bench_detail.json next to this file)')
    ap.add_argument('--detail_stdout', action='store_true',
                    help='also print the complete object as an earlier stdout line {"bench_detail": ...} (the last line stays compact)')
    ap.add_argument('--dry_launch', action='store_true',
                    help='launch-path check without a GPU: gloo group, count the ranks, print the JSON skeleton')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)                          # does not return

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus {} but WORLD_SIZE={}'.format(args.gpus, world))
    if args.dry_launch:
        return dry_launch(args, world, rank)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback)')
    # CMS_BENCH_SAME_DEVICE=1 (with CMS_BENCH_BACKEND=gloo): every rank on cuda:0 -- the only way to exercise the N > 1 code path
    # of this file on a 1-GPU box (numbers of such a run mean nothing; the driver never sets these)
    if os.environ.get('CMS_BENCH_SAME_DEVICE'):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # CMS_BENCH_FORCE_PG=1: bring the process group (RCCL) up even with ONE rank -- a 1-GPU box then measures the step with RCCL's
    # internal stream resident on one of the hardware queues, which is how every rank of the 8-GPU run executes it
    force_pg = world == 1 and os.environ.get('CMS_BENCH_FORCE_PG') == '1'
    if force_pg:
        os.environ.setdefault('MASTER_PORT', '29541')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    if world > 1 or force_pg:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(os.environ.get('CMS_BENCH_BACKEND', 'nccl'))
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)                     # RCCL really spans `world` ranks
        rccl_world = int(probe.item())
        # RCCL's internal stream exists from here on and sits on one of the four hardware queues: probe the side streams NOW so that
        # the step's roles (teacher | two weight-gradient streams) avoid it
        from cutmix_semisup_seg_amd import ops as _ops
        _ops.probe_streams(dev, again=True)
    else:
        rccl_world = 1

    if os.environ.get('CMS_BENCH_DUMMY_STREAMS'):      # EXPERIMENT: foreign streams created first (what RCCL / another library does)
        _dummy = [torch.cuda.Stream() for _ in range(int(os.environ['CMS_BENCH_DUMMY_STREAMS']))]
        for st_ in _dummy:
            with torch.cuda.stream(st_):
                torch.zeros(1, device=dev)
    keys = ['pascal', 'cityscapes'] if args.workload == 'both' else [args.workload]
    results = [run_workload(k, args, world, rank, dev) for k in keys]
    also = []
    if args.workload == 'both' and world == 1 and not args.no_also:
        # Two more configurations in the SAME driver-run line (short runs, never the headline `value`): BASELINE configs[3]
        # (DeepLab v3+ at 10x3x513x513) and configs[1] under the reference CLI's default BatchNorm mode (no --freeze_bn:
        # batch statistics). A failure here is reported in the entry and does not touch the main result.
        import argparse as _ap
        for name, key, over in (('pascal_v3plus (BASELINE configs[3])', 'pascal_v3plus', {}),
                                ('pascal without --freeze_bn (reference CLI default: batch-statistics BatchNorm)', 'pascal',
                                 {'no_freeze_bn': True})):
            a2 = _ap.Namespace(**vars(args))
            a2.steps, a2.warmup = min(args.steps, 10 if key == 'pascal_v3plus' else 20), min(args.warmup, 3)
            a2.no_cpu_baseline, a2.traffic, a2.timed_only = True, 'omit', True
            for k, v in over.items():
                setattr(a2, k, v)
            try:
                r = run_workload(key, a2, world, rank, dev)
                also.append({'name': name, 'value': r['value'], 'unit': 'images/sec', 'ms_per_step': r['ms_per_step'],
                             'steps': a2.steps, 'warmup': a2.warmup, 'config': r['config'],
                             'roofline': {k: r['roofline'].get(k) for k in ('kernel', 'frac', 'avg_launch_ms', 'step_mfma', 'mixed',
                                                                            'mixed_frac', 'step_mfma_frac')},
                             'roofline_hbm': ({k: r['roofline_hbm'].get(k) for k in ('group', 'frac', 'achieved', 'achieved_on_moved_bytes',
                                                                                    'ms_per_step')}
                                              if 'roofline_hbm' in r else None)})
            except Exception as e:                  # noqa: BLE001 -- reported, the headline line still goes out
                also.append({'name': name, 'error': '{}: {}'.format(type(e).__name__, e)})
                torch.cuda.empty_cache()
        # BASELINE configs[4]: VAT mean-teacher iteration of the DenseNet-161 U-Net (train_seg_semisup_vat_mt.py), 10 x 3 x 224 x 224,
        # 2 classes, SGD -- the gradient passes replayed as one hipGraph launch (vat.VATMeanTeacherStep); steady state after the capture
        try:
            also.append(vat_denseunet_run(dev))
        except Exception as e:                      # noqa: BLE001
            also.append({'name': 'VAT DenseNet-161 U-Net (BASELINE configs[4])', 'error': '{}: {}'.format(type(e).__name__, e)})
            torch.cuda.empty_cache()
    if rank == 0:
        head = results[0]
        out = {
            'metric': 'train images/sec (student+teacher step)',
            'value': head['value'],
            'unit': 'images/sec',
            'n_gpus': world,
            'rccl_world_size': rccl_world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': head['ms_per_step'],
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': args.dtype,
            'data': 'synthetic (N(0,1) images, uniform labels with 5% ignore=255, all-ones validity masks, '
                    'seeded box masks; random-init weights)',
            'config': head['config'],
            'roofline': head['roofline'],
        }
        for k in ('roofline_hbm', 'cpu_baseline'):
            if k in head:
                out[k] = head[k]
        for r in results:
            if r['workload'] == 'pascal':
                out['value_321x321'] = r['value']
            if r['workload'] == 'cityscapes':
                out['value_512x1024'] = r['value']
                # the 512 x 1024 shape of the metric as SCALAR keys of the objects the driver's record keeps
                out['config'] = dict(out['config'], value_512x1024=r['value'], ms_per_step_512x1024=r['ms_per_step'])
                rr = r['roofline']
                out['roofline'] = dict(out['roofline'], **{k + '_512x1024': rr[k] for k in
                                                           ('frac', 'mixed_frac', 'step_mfma_frac', 'isolated_frac', 'hbm_group_frac',
                                                            'hbm_group_ms_per_step', 'avg_launch_ms') if rr.get(k) is not None})
        out['configs'] = results
        if also:
            out['also'] = also
            for a_, k_ in zip(also, ('also_v3plus_513x513_img_s', 'also_no_freeze_bn_321x321_img_s', 'also_vat_denseunet_224x224_img_s')):
                out['config'][k_] = a_.get('value')
        # the complete object goes to bench_detail.json (and, with --detail_stdout, to an EARLIER stdout line); the LAST stdout
        # line is the compact one the driver parses (<= LINE_LIMIT bytes)
        dpath = write_detail(out, args.detail_path or None)
        if dpath:
            out['detail'] = os.path.basename(dpath)
        if args.detail_stdout:
            print(json.dumps({'bench_detail': out}))
        sys.stdout.flush()
        print(compact_line(out))
    if world > 1 or force_pg:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
