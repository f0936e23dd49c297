"""
GPU: DeepLab v2 WITHOUT --freeze_bn -- the reference CLI's default (train_seg_semisup_mask_mt.py:587): every BatchNorm
normalises with batch statistics and moves its running statistics, in the student AND in the train-mode teacher (Q4), while
its affine parameters stay frozen (deeplab2.py:72-84, Q3). One whole CutMix mean-teacher iteration on the hand-written
kernels against oracle/step.py with frozen_bn=False: losses, confidence rate, every gradient, student and teacher running
statistics -- on BOTH routes of this build:
  * 'executor': body and head on the static executor, every unit = convolution + csrc/bn.hip launches recorded into the
    same programs (cms_program_add_bn); the stem through the strict layer engine (7 x 7 as tap chunks);
  * 'layers': everything through the strict layer engine (what runs under torch.distributed, where BatchNorm all-reduces
    its statistics between its two passes).
The `no_library_convolutions` context refuses any library convolution / BatchNorm call in either.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _setup(dtype, kind, C=5, layers=(1, 1, 1, 1), lr=1e-3):
    from architectures import deeplab2
    from cutmix_semisup_seg_amd import optim as fo
    from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig
    from oracle import deeplab2 as odl
    import optim_weight_ema
    layers = list(layers)
    # seeded He initialisation: the closed-form fixture weights make deep gradients cancel by orders of magnitude (what is
    # left is rounding noise, cf. tests/test_gpu_deeplab3plus.py), useless where gradients are compared
    g = torch.Generator().manual_seed(77)
    st = {}
    for k, (shape, dt) in odl.state_spec(C, layers).items():
        if dt == torch.int64:
            st[k] = torch.zeros(shape, dtype=torch.int64)
        elif len(shape) == 4:
            st[k] = torch.randn(shape, generator=g) * (2.0 / (shape[1] * shape[2] * shape[3])) ** 0.5 * (0.3 if k.startswith('layer5.') else 1.0)
        elif k.endswith('running_var'):
            st[k] = 0.8 + 0.4 * torch.rand(shape, generator=g)
        elif k.endswith('running_mean'):
            st[k] = 0.1 * torch.randn(shape, generator=g)
        elif k.endswith('.weight'):
            st[k] = 0.6 + 0.8 * torch.rand(shape, generator=g)
        else:
            st[k] = 0.1 * torch.randn(shape, generator=g)
    mk = lambda: deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    stu, tea = mk(), mk()
    stu.load_state_dict(st)
    stu, tea = stu.to(DEV), tea.to(DEV)
    stu.compute_dtype = tea.compute_dtype = dtype
    stu.engine_kind = tea.engine_kind = kind
    opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=lr * 0.1),
                             dict(params=list(stu.new_parameters()), lr=lr)])
    for p in tea.parameters():
        p.requires_grad = False
    ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
    ema.fuse_into(opt)
    stu.train(); tea.train()                          # NO freeze_batchnorm(): batch statistics everywhere
    step = CutMixMeanTeacherStep(stu, tea, opt, ema, StepConfig(conf_thresh=0.2, compute_dtype=dtype))
    return st, stu, tea, opt, step


def _rel(a, b):
    return float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize('route', ['executor', 'executor_separate_passes', 'layers'])
def test_batch_statistics_iteration_on_the_hand_written_kernels_matches_the_oracle(no_library_convolutions, route):
    from cutmix_semisup_seg_amd import ops
    from cutmix_semisup_seg_amd.step import UnsupBatch
    from oracle import step as ostep, boxmask as obox
    import mask_gen
    C, layers, N, H, W = 5, [1, 1, 1, 1], 3, 49, 65
    st, stu, tea, opt, step = _setup(torch.float32, 'hip', C, layers)
    if route == 'layers':
        stu.batchstat_executor = tea.batchstat_executor = False
    if route == 'executor_separate_passes':
        step.cfg.fuse_batches = False
    assert not step._samples_independent() and stu._use_hip_body() == (route != 'layers')
    # 'executor': the four forward passes of the reference travel as two grouped batches ([sup; mixed] through the student,
    # [x0; x1] through the teacher), the BatchNorm kernels keeping the groups' statistics apart (step._sample_groups)
    assert (step._sample_groups(N, [None], False) is not None) == (route != 'layers')
    g = torch.Generator().manual_seed(21)
    x, ux0, ux1 = (torch.randn(N, 3, H, W, generator=g) for _ in range(3))
    y = torch.randint(0, C, (N, 1, H, W), generator=g)
    y[torch.rand(N, 1, H, W, generator=g) < 0.05] = 255
    ranges = mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(N, (H, W), rng=np.random.RandomState(3))
    m = torch.tensor(obox.rasterise(ranges, (H, W), True).astype(np.float32))
    ones = torch.ones(N, 1, H, W)
    S = ostep.StepState(st, C, layers, opt='adam', lr=1e-3, teacher_alpha=0.99)
    grads = {}
    ref = ostep.train_iteration(S, x, y, ux0, ux1, ones, ones, m, conf_thresh=0.2, frozen_bn=False, grads_out=grads)
    tea_before = {k: v.clone() for k, v in tea.state_dict().items()}
    with no_library_convolutions:
        res = step(x.to(DEV), y.to(torch.uint8).to(DEV), [UnsupBatch(ux0.to(DEV), ops.ranges_to_device(ranges, DEV),
                                                                    x1_tea=ux1.to(DEV))])
    got = {k: float(v) for k, v in res.items()}
    assert no_library_convolutions.refused == 0
    eng = stu._hip_engine
    assert eng is not None and eng.strict and eng.dtype == torch.float32 and eng.library_convs == 0
    if route != 'layers':
        progs = stu._hip_executor.programs()
        shapes = sorted(int(p.x_in.shape[0]) for p in progs if hasattr(p, 'x_in'))
        assert shapes == ([2 * N] if route == 'executor' else [N]), shapes
        assert stu._hip_executor.dtype == torch.float32 and len(progs) >= 2 and all(getattr(p, 'bn', True) for p in progs if hasattr(p, 'bn'))
        assert tea._hip_executor is not None
    else:
        assert stu._hip_executor is None
    assert abs(got['sup_loss'] - ref['sup_loss']) <= 1e-4 * abs(ref['sup_loss'])
    assert abs(got['consistency_loss'] - ref['consistency_loss']) <= 2e-3 * abs(ref['consistency_loss']) + 1e-9
    assert abs(got['conf_rate'] - ref['conf_rate']) <= 2e-3
    rels = {k: _rel(p.grad, grads[k]) for k, p in stu.named_parameters() if grads.get(k) is not None}
    assert len(rels) >= 15
    worst = sorted(rels.items(), key=lambda kv: -kv[1])[:4]
    print('\nPARITY DeepLab v2 batch-statistics iteration (fp32, hand-written kernels) vs oracle: got {} ref {} gradients max '
          '{:.2e} mean {:.2e} worst {} all {}'.format(got, ref, max(rels.values()), float(np.mean(list(rels.values()))), worst,
                                                     {k: float('{:.1e}'.format(v)) for k, v in rels.items()}))
    # Every operator of this pass is exact on its own (tools/debug_bn_layers.py, debug_conv_layers.py: each BatchNorm and each
    # convolution, forward / data gradient / weight gradient, within 1e-7 .. 8e-7 of fp64 on the device's own inputs). What
    # separates device and CPU oracle end to end is ONE ReLU whose pre-activation is zero to rounding and lands on the other
    # side (tools/debug_du_layers.py: a single sign mismatch in layer4.0's first activation): with batch statistics over only
    # 3 x 7 x 9 positions that one element moves every gradient BELOW it by ~3e-3 (1 / sqrt(#elements)); everything above
    # it -- head, layer4.0 conv2 / conv3 / downsample -- agrees to 4e-6. Asserted accordingly.
    above = [k for k in rels if k.startswith('layer5.') or k in ('layer4.0.conv2.weight', 'layer4.0.conv3.weight',
                                                                 'layer4.0.downsample.0.weight')]
    assert max(rels[k] for k in above) <= 2e-5, {k: rels[k] for k in above}
    assert max(rels.values()) <= 1.5e-2 and float(np.mean(list(rels.values()))) <= 6e-3, worst
    # running statistics: the student saw two passes; the teacher two passes and then the EMA blend with the student's
    sd_s, sd_t = stu.state_dict(), tea.state_dict()
    for k in sd_s:
        if k.endswith('running_mean') or k.endswith('running_var'):
            assert float((sd_s[k].cpu() - S.student[k]).abs().max()) <= 1e-4 * float(S.student[k].abs().max()) + 1e-6, k
            assert float((sd_t[k].cpu() - S.teacher[k]).abs().max()) <= 1e-4 * float(S.teacher[k].abs().max()) + 1e-6, k
            assert float((sd_t[k].cpu() - tea_before[k].cpu()).abs().max()) > 0, k
    assert int(sd_s['bn1.num_batches_tracked']) == 2


def test_bf16_batch_statistics_iteration_runs_on_the_mfma_engine():
    """The default ('auto') engine in bf16: eligible convolutions on csrc/conv.hip, BatchNorm on csrc/bn.hip; the loss falls."""
    from cutmix_semisup_seg_amd import ops
    from cutmix_semisup_seg_amd.step import UnsupBatch
    from cutmix_semisup_seg_amd.architectures.deeplab3plus import HipConvEngine
    import mask_gen
    C, N, H, W = 5, 4, 97, 97
    st, stu, tea, opt, step = _setup(torch.bfloat16, 'auto', C, (1, 1, 2, 1), lr=1e-4)
    g = torch.Generator(device=DEV).manual_seed(1)
    y = (torch.rand(N, 1, H, W, generator=g, device=DEV) * C).long().clamp_(0, C - 1).to(torch.uint8)
    x = (torch.randn(N, 3, H, W, generator=g, device=DEV) + 0.5 * y.float()).bfloat16()
    im = lambda: torch.randn(N, 3, H, W, generator=g, device=DEV).bfloat16()
    ranges = ops.ranges_to_device(mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(
        N, (H, W), rng=np.random.RandomState(0)), DEV)
    losses = [float(step(x, y, [UnsupBatch(im(), ranges, x1_tea=im())])['sup_loss']) for _ in range(20)]
    print('\nbf16 batch-statistics DeepLab v2 losses:', [round(v, 4) for v in losses])
    assert isinstance(stu._hip_engine, HipConvEngine)            # (the stem's layer engine; body + head on the executor)
    assert stu._hip_executor is not None and stu._hip_executor.batch_statistics()
    assert all(np.isfinite(losses)) and min(losses[-5:]) < losses[0]
