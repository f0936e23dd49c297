"""
GPU: the hand-written MFMA executor of the DeepLab v2 body (backbone_hip.py: fused conv/BN/ReLU/residual forward,
dgrad/wgrad backward) against the library engine (same network, same bf16 weights) and the fp32 CPU oracle.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _cf_input(n, h, w, phase):
    idx = torch.arange(n * 3 * h * w, dtype=torch.float64)
    return torch.sin(phase + 0.61803398875 * idx).reshape(n, 3, h, w).float() * 1.5


def _state(layers, C, active_relus):
    """
    Seeded random weights with a healthy gradient flow (He-initialised convolutions, BN scale in [0.6, 1.4], small
    running statistics). With `active_relus` every BN bias is pushed up so that (almost) all ReLUs are in their linear
    region -- the backward pass then has no ReLU-mask ambiguity between bf16 and fp32 activations.
    (The closed-form weights of the golden fixtures make early-layer gradients cancel by 2-3 orders of magnitude per
    stage, which turns any bf16 backward -- library or hand-written -- into noise; not a useful yardstick here.)
    """
    from oracle import deeplab2 as odl
    g = torch.Generator().manual_seed(1234)
    st = {}
    for k, (shape, dt) in odl.state_spec(C, layers).items():
        if dt == torch.int64:
            st[k] = torch.zeros(shape, dtype=torch.int64)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            st[k] = torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
        elif k.endswith('running_var'):
            st[k] = 0.8 + 0.4 * torch.rand(shape, generator=g)
        elif k.endswith('running_mean'):
            st[k] = 0.1 * torch.randn(shape, generator=g)
        elif k.endswith('.weight'):
            st[k] = 0.6 + 0.8 * torch.rand(shape, generator=g)
        else:
            st[k] = 0.1 * torch.randn(shape, generator=g) + (1.5 if (active_relus and not k.startswith('layer5')) else 0.0)
    return st


def _build(layers, C, kind, active_relus=False, dtype=torch.bfloat16):
    """kind 'library': the MIOpen comparison engine of tests/_library_engine.py plugged in as the network's engine object."""
    from architectures import deeplab2
    net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    net.load_state_dict(_state(layers, C, active_relus))
    net = net.to(DEV)
    net.compute_dtype = dtype
    if kind == 'library':
        from _library_engine import LibraryEngine
        net.engine = LibraryEngine(dtype)
    else:
        net.engine_kind = kind
    net.train()
    net.freeze_batchnorm()
    return net


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize('active', [True, False], ids=['linear_regime', 'relu_regime'])
@pytest.mark.parametrize('layers,C,shape', [([1, 1, 1, 1], 5, (2, 65, 81)), ([3, 4, 23, 3], 21, (2, 65, 65)),
                                            ([3, 4, 23, 3], 19, (1, 97, 129))], ids=['tiny', 'r101_21', 'r101_19'])
def test_executor_forward_backward_matches_library_engine(layers, C, shape, active):
    from oracle import deeplab2 as odl
    n, H, W = shape
    x = _cf_input(n, H, W, 0.7).bfloat16().to(DEV)
    hip, ref = _build(layers, C, 'hip', active), _build(layers, C, 'library', active)
    lo_h = hip.forward_lowres(x)
    lo_r = ref.forward_lowres(x)
    assert lo_h.shape == lo_r.shape and lo_h.dtype == torch.float32 and lo_h.is_contiguous()
    # fp32 CPU oracle on the same bf16-rounded conv weights: the hand-written path (one rounding per layer) must be at
    # least as close to it as the library path (several roundings per layer)
    st = _state(layers, C, active)
    st_bf = {k: (v.bfloat16().float() if (v.dtype == torch.float32 and v.dim() == 4) else v) for k, v in st.items()}
    want = odl.forward_lowres(x.float().cpu(), st_bf, layers, frozen=True)
    e_h, e_r = _rel(lo_h.cpu(), want), _rel(lo_r.cpu(), want)
    assert e_h <= 6e-2, (e_h, e_r)
    assert e_h <= 1.5 * e_r + 5e-3, (e_h, e_r)
    # backward with the same upstream gradient; truth = the fp32 LIBRARY engine (an independent implementation); the bf16 library
    # engine's error is printed beside it for orientation only
    ref32 = _build(layers, C, 'library', active, dtype=torch.float32)
    lo_32 = ref32.forward_lowres(x)
    g = torch.randn(lo_h.shape, generator=torch.Generator(device=DEV).manual_seed(1), device=DEV)
    hip._cms_arena.zero_grad()
    lo_h.backward(g)
    lo_r.backward(g)
    lo_32.backward(g)
    named_h, named_r, named_t = dict(hip.named_parameters()), dict(ref.named_parameters()), dict(ref32.named_parameters())
    keys = ['conv1.weight', 'layer1.0.conv1.weight', 'layer1.0.conv2.weight', 'layer1.0.downsample.0.weight',
            'layer2.0.conv1.weight', 'layer2.0.downsample.0.weight', 'layer3.0.conv2.weight', 'layer4.0.conv3.weight',
            'layer5.conv2d_list.0.weight', 'layer5.conv2d_list.1.weight', 'layer5.conv2d_list.0.bias']
    errs = {}
    for k in keys:
        gh, gr, gt = named_h[k].grad, named_r[k].grad, named_t[k].grad
        assert gh is not None and gr is not None, k
        errs[k] = (round(_rel(gh, gt), 4), round(_rel(gr.float(), gt), 4))
    print('gradient rel. errors vs fp32 (hand-written, library bf16):', errs)
    # bf16 activations / gradients through up to 101 layers deviate from the fp32 gradients by a few % at the head and tens of % at
    # the stem -- storage noise (the per-layer teacher-forced test is the tight statement). ADVICE r5: the bound is
    # ABSOLUTE per layer group, not a multiple of the library's bf16 error (a yardstick that moved between 0.15 and 0.27 from process
    # to process with MIOpen's algorithm choice while the hand-written value stayed at 0.2951). Hand-written values measured over
    # rounds 2-5: head <= 0.0095, layer4.0.conv3 <= 0.101, body <= 0.137 (tiny) / <= 0.352 (ResNet-101).
    body_cap = 0.20 if sum(layers) == 4 else 0.45
    body = [k for k in keys if not k.startswith('layer5.') and k != 'layer4.0.conv3.weight']
    assert all(errs[k][0] <= body_cap for k in body), errs
    assert float(np.mean([errs[k][0] for k in body])) <= 0.75 * body_cap, errs
    assert errs['layer4.0.conv3.weight'][0] <= 0.15, errs
    assert all(errs[k][0] <= 2e-2 for k in keys if k.startswith('layer5.')), errs
    assert float(named_h['layer5.conv2d_list.2.weight'].grad.abs().max()) == 0.0     # never receives a gradient


def test_executor_teacher_forward_only_and_weight_refresh():
    from oracle import deeplab2 as odl
    net = _build([1, 1, 1, 1], 5, 'hip')
    for p in net.parameters():
        p.requires_grad = False
    x = _cf_input(1, 33, 47, 0.1).bfloat16().to(DEV)
    with torch.no_grad():
        a = net.forward_lowres(x)
        st = odl.closed_form_state(5, [1, 1, 1, 1])
        st2 = {k: (v * 0.5 if k == 'layer5.conv2d_list.0.weight' else v) for k, v in st.items()}
        net.load_state_dict(st2)           # the post-hook refreshes the bf16 operands
        b = net.forward_lowres(x)
    assert float((a - b).abs().max()) > 1e-3
    assert net._hip_executor.arena.grad is None


@pytest.mark.parametrize('route', ['executor', 'layers'])
def test_hip_engine_without_frozen_bn_stays_on_the_hand_written_kernels(no_library_convolutions, route):
    """Round 2 refused this configuration (the static executor folded FROZEN BatchNorm into its epilogues). Since round 3
    batch-statistics passes -- the reference CLI's default, no --freeze_bn -- run on the executor with csrc/bn.hip launches
    recorded into its programs (stem through the strict layer engine), or, with `batchstat_executor = False` (and always
    under torch.distributed), layer by layer through the strict engine. Never the library."""
    from architectures import deeplab2
    from cutmix_semisup_seg_amd.architectures.deeplab3plus import HipConvEngine
    net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, [1, 1, 1, 1], 5, np.zeros(3), np.ones(3)).to(DEV)
    net.engine_kind = 'hip'
    if route == 'layers':
        net.batchstat_executor = False
    net.train()
    assert net._use_hip_body() == (route == 'executor')
    with no_library_convolutions:
        out = net.forward_lowres(torch.randn(2, 3, 33, 33, device=DEV).bfloat16())
        out.sum().backward()
    assert isinstance(net._hip_engine, HipConvEngine) and net._hip_engine.strict and net._hip_engine.library_convs == 0
    assert tuple(out.shape) == (2, 5, 5, 5) and bool(torch.isfinite(out).all()) and no_library_convolutions.refused == 0
    if route == 'executor':
        assert net._hip_executor is not None and net._hip_executor.batch_statistics() and len(net._hip_executor.programs()) >= 2
    else:
        assert net._hip_executor is None
    assert int(net.bn1.num_batches_tracked) == 1 and int(net.layer3[0].bn2.num_batches_tracked) == 1
    assert all(p.grad is not None and float(p.grad.abs().max()) > 0 for k, p in net.named_parameters()
               if p.requires_grad and 'conv2d_list.2' not in k and 'conv2d_list.3' not in k)
    net.freeze_batchnorm()
    assert net._use_hip_body() and not net._hip_executors[net.compute_dtype].batch_statistics() if route == 'executor' else True


def test_whole_step_bf16_hip_engine_tracks_fp32_library_engine():
    """Three iterations: bf16 / MFMA executor vs fp32 / library engine on identical data."""
    from architectures import deeplab2
    from oracle import deeplab2 as odl
    import mask_gen
    import optim_weight_ema
    from cutmix_semisup_seg_amd import ops, optim as fo
    from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig, UnsupBatch
    C, layers, N, H, W = 5, [1, 1, 1, 1], 2, 65, 65
    logs = {}
    for tag, dtype in (('hip', torch.bfloat16), ('lib', torch.float32)):      # 'lib': fp32 parity configuration
        mk = lambda: deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
        stu, tea = mk(), mk()
        stu.load_state_dict(odl.closed_form_state(C, layers))
        stu, tea = stu.to(DEV), tea.to(DEV)
        stu.compute_dtype = tea.compute_dtype = dtype
        opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=1e-4),
                                 dict(params=list(stu.new_parameters()), lr=1e-3)])
        for p in tea.parameters():
            p.requires_grad = False
        ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
        ema.fuse_into(opt)
        stu.train(); tea.train(); stu.freeze_batchnorm(); tea.freeze_batchnorm()
        step = CutMixMeanTeacherStep(stu, tea, opt, ema, StepConfig(conf_thresh=0.3, compute_dtype=dtype))
        gen_m = mask_gen.BoxMaskGenerator(0.5, invert=True)
        rng = np.random.RandomState(3)
        out = []
        for it in range(3):
            g = torch.Generator().manual_seed(50 + it)
            x = torch.randn(N, 3, H, W, generator=g).to(DEV).to(dtype)
            y = torch.randint(0, C, (N, 1, H, W), generator=g).to(torch.uint8).to(DEV)
            ux0 = torch.randn(N, 3, H, W, generator=g).to(DEV).to(dtype)
            ux1 = torch.randn(N, 3, H, W, generator=g).to(DEV).to(dtype)
            ranges = ops.ranges_to_device(gen_m.generate_ranges(N, (H, W), rng=rng), DEV)
            r = step(x, y, [UnsupBatch(ux0, ranges, x1_tea=ux1)])
            out.append([float(r['sup_loss']), float(r['consistency_loss']), float(r['conf_rate'])])
        logs[tag] = np.array(out)
        if tag == 'hip':
            assert stu._hip_executor is not None and tea._hip_executor is not None
            w_hip = stu.state_dict()['layer3.0.conv2.weight'].float().cpu()
        else:
            w_lib = stu.state_dict()['layer3.0.conv2.weight'].float().cpu()
    # three Adam steps at lr 1e-3 on the closed-form weights: the two trajectories drift apart step by step (sign-like
    # first updates amplify gradient noise); the one-iteration comparison against the oracle with healthy weights is
    # tests/test_gpu_hip_engine_parity.py
    np.testing.assert_allclose(logs['hip'][:2, 0], logs['lib'][:2, 0], rtol=8e-2)
    np.testing.assert_allclose(logs['hip'][2:, 0], logs['lib'][2:, 0], rtol=0.2)
    np.testing.assert_allclose(logs['hip'][:, 1], logs['lib'][:, 1], rtol=0.25, atol=1e-6)
    np.testing.assert_allclose(logs['hip'][:, 2], logs['lib'][:, 2], atol=3e-2)
    # Adam takes lr-sized steps: after 3 iterations the two weight sets moved the same way
    st0 = odl.closed_form_state(C, layers)['layer3.0.conv2.weight']
    dh, dl = (w_hip - st0).flatten(), (w_lib - st0).flatten()
    cos = float((dh * dl).sum() / (dh.norm() * dl.norm() + 1e-30))
    assert cos > 0.9, cos


@pytest.mark.parametrize('arch,crop', [('resnet101_deeplab_imagenet', '65,65'),
                                       ('resnet101_deeplabv3plus_imagenet', '129,129')])
def test_trainer_cli_synthetic_end_to_end(tmp_path, monkeypatch, capsys, arch, crop):
    """The reference's command line (run_pascal_aug_experiments.sh / ..._deeplab3plus_experiments.sh flag sets) on
    synthetic data: 2 epochs x 2 iterations of the ResNet-101 networks at a small crop; checks the job/log layout and
    the epoch log line format."""
    import re
    from click.testing import CliRunner
    import train_seg_semisup_mask_mt as trainer
    monkeypatch.chdir(tmp_path)
    args = ['--job_desc', 'smoke', '--synthetic', '--arch', arch, '--freeze_bn', '--batch_size', '2',
            '--crop_size', crop, '--learning_rate', '3e-5', '--lr_sched', 'poly', '--mask_prop_range', '0.5',
            '--conf_thresh', '0.97', '--num_epochs', '2', '--iters_per_epoch', '2', '--synthetic_val_batches', '1']
    res = CliRunner().invoke(trainer.experiment, args, catch_exceptions=False)
    assert res.exit_code == 0, res.output
    log = open(tmp_path / 'results' / 'train_seg_semisup_mask_mt' / 'log_smoke.txt').read()
    lines = [l for l in log.splitlines() if l.startswith('Epoch ')]
    assert len(lines) == 2
    pat = re.compile(r'Epoch \d+: took [\d.]+s, TRAIN clf loss=[\d.]+, consistency loss=[\d.]+, conf rate=[\d.]+%, '
                     r'VAL mIoU=[\d.]+%')
    assert all(pat.match(l) for l in lines), lines
    assert 'Built network' in log and 'Training...' in log and 'Settings:' in log
    clf = [float(re.search(r'clf loss=([\d.]+)', l).group(1)) for l in lines]
    assert all(np.isfinite(v) and 0 < v < 10 for v in clf)


def test_eval_and_full_resolution_forward_with_hip_engine():
    import evaluation
    from cutmix_semisup_seg_amd import ops
    net = _build([1, 1, 1, 1], 5, 'hip')
    net.eval()
    x = _cf_input(2, 65, 65, 0.2).bfloat16().to(DEV)
    with torch.no_grad():
        lo = net.forward_lowres(x)
        full = net(x)                                   # the reference's forward contract: logits at input size
    assert full.shape == (2, 5, 65, 65)
    torch.testing.assert_close(full, ops.upsample_bilinear(lo, (65, 65), True))
    y = torch.randint(0, 5, (2, 1, 65, 65), device=DEV).to(torch.uint8)
    ev = evaluation.EvaluatorIoU(5)
    ev.sample_logits(lo, y, (65, 65), ignore_value=255, align_corners=True)
    ev2 = evaluation.EvaluatorIoU(5)
    pred = full.argmax(dim=1)
    for i in range(2):
        ev2.sample(y[i, 0], pred[i].to(torch.uint8), ignore_value=255)
    assert abs(ev.score().mean() - ev2.score().mean()) < 2e-3     # fused vs materialised path (fp ties aside)


def test_frozen_batchnorm_fold_kernel_equals_the_tensor_expression():
    """cms_bn_fold (round 5: one launch for all layers of a network instead of nine tensor ops): scale = gamma * rsqrt(var + eps),
    bias = beta - mean * scale (deeplab2.py:92-107 with frozen statistics) -- bit for bit the tensor expression evaluated on the
    CPU (correctly rounded 1 / sqrt: what oracle/deeplab2_chain.py folds with), within an ulp of the GPU library's rsqrt."""
    from cutmix_semisup_seg_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    flat = torch.randn(50000, generator=g, device=DEV)
    n = 7777
    ix = {k: torch.randint(0, 50000, (n,), generator=g, device=DEV) for k in ('w', 'b', 'm', 'v')}
    flat[ix['v']] = flat[ix['v']].abs() + 1e-3                       # variances
    scale, bias = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    ops.bn_fold(flat, ix['w'], ix['b'], ix['m'], ix['v'], 1e-5, scale, bias)
    fc, ic = flat.cpu(), {k: v.cpu() for k, v in ix.items()}
    want_s = fc[ic['w']] * torch.rsqrt(fc[ic['v']] + 1e-5)
    want_b = fc[ic['b']] - fc[ic['m']] * want_s
    assert torch.equal(scale.cpu(), want_s) and torch.equal(bias.cpu(), want_b)
    lib_s = flat[ix['w']] * torch.rsqrt(flat[ix['v']] + 1e-5)                     # the expression of rounds 1-4, on the device
    assert float(((scale - lib_s).abs() / lib_s.abs().clamp_min(1e-30)).max()) <= 2.5e-7
    # ... and through the executor: the folded affine of a real network
    net = _build([1, 1, 1, 1], 5, 'hip')
    ex = net.hip_executor()
    ex._refresh_affine()
    bn = net.layer3[0].bn2
    w_, b_, m_, v_ = (t.detach().cpu() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
    s = w_ * torch.rsqrt(v_ + 1e-5)
    c = ex.blocks[ex.layer_first_blocks()[2]].c2
    assert torch.equal(c.scale.cpu(), s) and torch.equal(c.bias.cpu(), b_ - m_ * s)


def test_side_streams_are_probed_to_overlap_with_the_default_stream_and_each_other():
    """Round 5: `ops.pooled_stream` hands out streams from a PROBED set -- the runtime maps streams onto a few hardware queues and two
    streams on one queue run in order (a teacher / weight-gradient stream on the main stream's queue costs the step 20-45 %,
    profiles/r05k_*). The chosen streams must overlap pairwise: two ~0.1 ms spin kernels side by side take about as long as one."""
    from cutmix_semisup_seg_amd import ops
    dev = torch.device(DEV)
    tea, w0, w1 = (ops.pooled_stream(dev, r) for r in ('teacher', 'wgrad0', 'wgrad1'))
    assert ops._STREAM_PROBE_LOG and len(ops._STREAM_PROBE_LOG[0][3]) >= 2
    assert ops.pooled_stream(dev, 'optimizer') is w1                       # default alias: the early optimizer launch behind wgrad1
    cur = torch.cuda.current_stream()
    spin = 200000

    def pair_ms(a, b):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(a)
        with torch.cuda.stream(a):
            torch.cuda._sleep(spin)
        if b is not None:
            b.wait_event(e0)
            with torch.cuda.stream(b):
                torch.cuda._sleep(spin)
            a.wait_stream(b)
        e1.record(a)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    pair_ms(cur, tea)
    # The spin kernel counts SHADER clocks: its duration moves with the power state (0.090 ms at 2.2 GHz, 0.150 ms on a GPU that has
    # just idled -- which read as "same queue" against an `alone` measured a moment earlier at full clock, round 6). Every trial
    # therefore measures the single kernel and the pair back to back and compares THEM; the best of a few trials counts (noise --
    # another process on the GPU, a clock step between the two halves of a trial -- only ever raises a ratio).
    def ratio(a, b):
        torch.cuda._sleep(20 * spin)                      # ~2 ms of work first: the clock is up when the trial starts
        r = []
        for _ in range(6):
            alone = pair_ms(a, None)
            r.append(pair_ms(a, b) / max(alone, 1e-6))
        return min(r)
    streams = [cur, tea, w0, w1]
    distinct = len({int(s.cuda_stream) for s in streams})
    if distinct == 4:
        for i in range(4):
            for j in range(i + 1, 4):
                rij = ratio(streams[i], streams[j])
                assert rij < 1.6, (i, j, rij)
