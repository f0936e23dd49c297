"""
GPU: run-to-run determinism of the weight gradients. By default the split-K partial sums of
csrc/conv.hip's weight-gradient kernel are combined with fp32 atomics, whose order varies from run to run (the throughput
default: 1.5-2 % faster inside the two-stream step). `StepConfig(deterministic=True)` (trainers: --deterministic) routes them
through per-launch slabs and an ordered reduce: two runs of the same iteration must then agree BIT FOR BIT on every gradient
tensor, and the atomics path is shown to differ (so the test would notice a silently ignored switch).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _one_iteration(deterministic, seed=3, freeze_bn=True, early_optimizer=True, iters=1):
    from cutmix_semisup_seg_amd import ops, optim as fo
    from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig, UnsupBatch
    from architectures import deeplab2
    import mask_gen
    import optim_weight_ema
    ops.set_deterministic_wgrad(False)
    C, layers, N, H, W = 7, [2, 2, 3, 2], 4, 161, 161
    torch.manual_seed(seed)
    mk = lambda: deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3)).to(DEV)
    stu, tea = mk(), mk()
    opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=1e-5),
                             dict(params=list(stu.new_parameters()), lr=1e-4)])
    for p in tea.parameters():
        p.requires_grad = False
    ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
    ema.fuse_into(opt)
    stu.train(); tea.train()
    if freeze_bn:
        stu.freeze_batchnorm(); tea.freeze_batchnorm()
    stu.engine_kind = tea.engine_kind = 'hip'
    step = CutMixMeanTeacherStep(stu, tea, opt, ema, StepConfig(conf_thresh=0.2, deterministic=deterministic,
                                                              early_optimizer=early_optimizer))
    g = torch.Generator(device=DEV).manual_seed(11)
    im = lambda: torch.randn(N, 3, H, W, generator=g, device=DEV).bfloat16()
    y = torch.randint(0, C, (N, 1, H, W), generator=g, device=DEV).to(torch.uint8)
    r = mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(N, (H, W), rng=np.random.RandomState(5))
    for _ in range(iters):
        res = step(im(), y, [UnsupBatch(im(), ops.ranges_to_device(r, DEV), x1_tea=im())])
    torch.cuda.synchronize()
    _one_iteration.last_state = ({k: v.clone() for k, v in stu.state_dict().items()},
                                 {k: v.clone() for k, v in tea.state_dict().items()}, int(opt.step_count.item()))
    grads = {k: p.grad.detach().clone() for k, p in stu.named_parameters() if p.grad is not None}
    ops.set_deterministic_wgrad(False)
    return {k: float(v) for k, v in res.items()}, grads, stu.state_dict()['layer3.1.conv2.weight'].clone()


def test_deterministic_mode_gives_bit_identical_gradients_run_to_run():
    ra, ga, wa = _one_iteration(True)
    rb, gb, wb = _one_iteration(True)
    assert ra == rb
    differing = [k for k in ga if not torch.equal(ga[k], gb[k])]
    print('\ndeterministic mode: {} of {} gradient tensors differ between two runs: {}'.format(len(differing), len(ga), differing[:8]))
    assert not differing
    assert torch.equal(wa, wb)                                   # ... and so does the updated weight
    # the default (atomics) path is NOT bit-reproducible -- and the values agree to fp32 summation noise
    _, gc, _ = _one_iteration(False)
    _, gd, _ = _one_iteration(False)
    n_diff = sum(1 for k in gc if not torch.equal(gc[k], gd[k]))
    rel = max(float((ga[k] - gc[k]).norm() / (ga[k].norm() + 1e-30)) for k in ga)
    print('atomics mode: {} of {} tensors differ run to run; deterministic vs atomics max rel {:.2e}'.format(n_diff, len(gc), rel))
    assert rel <= 1e-4


def test_deterministic_mode_with_batch_statistics_is_bit_reproducible_too():
    """The reference CLI's default BatchNorm mode (no --freeze_bn): the statistics reductions of csrc/bn.hip add their partial
    sums in a fixed order (no data atomics since round 3), so with the deterministic weight gradients the whole grouped
    iteration -- losses, every gradient, the updated weights, the running statistics -- repeats bit for bit."""
    ra, ga, wa = _one_iteration(True, freeze_bn=False)
    rb, gb, wb = _one_iteration(True, freeze_bn=False)
    assert ra == rb
    differing = [k for k in ga if not torch.equal(ga[k], gb[k])]
    print('\nbatch-statistics, deterministic mode: {} of {} gradient tensors differ between two runs: {}'.format(
        len(differing), len(ga), differing[:8]))
    assert not differing and torch.equal(wa, wb)


def test_early_optimizer_launches_change_nothing():
    """The optimizer + EMA kernel issued per finished gradient slice during the backward pass (StepConfig.early_optimizer, the
    default) against ONE launch after it: in deterministic mode student, teacher and step counter agree bit for bit after
    three iterations."""
    _one_iteration(True, iters=3, early_optimizer=True)
    sa, ta, na = _one_iteration.last_state
    _one_iteration(True, iters=3, early_optimizer=False)
    sb, tb, nb = _one_iteration.last_state
    assert na == nb == 3
    bad = [k for k in sa if not torch.equal(sa[k], sb[k])] + ['teacher.' + k for k in ta if not torch.equal(ta[k], tb[k])]
    assert not bad, bad[:8]
