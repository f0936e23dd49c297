#!/usr/bin/env python
"""Run-to-run reproducibility of ONE training step in one process: the same step on four freshly built, identically seeded
student / teacher pairs (deterministic weight gradients on) -- losses, the largest gradient difference, the first bottleneck of
the backward chain whose data gradients differ, the weight-gradient tensors that differ.
    python tools/step_repro_check.py [arch]          DBG_HIP=1: the all-hand-written engine (engine_kind = 'hip')
Round 4: DeepLab v2 0.0; DeepLab v3+ on the 'auto' engine 5e-3 ... 1e-2 of the gradients (the library convolution of the
pooled branch varies in its last bits, bf16 storage amplifies it), 1e-7 on the 'hip' engine (atomics of the BatchNorm-affine
side outputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cutmix_semisup_seg_amd import ops, optim as fo
from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig, UnsupBatch
from architectures import network_architectures
import mask_gen, optim_weight_ema
dev = torch.device('cuda:0')
C, N, H, W = 5, 2, 65, 65
arch = sys.argv[1] if len(sys.argv) > 1 else 'resnet101_deeplabv3plus_imagenet'
def build():
    torch.manual_seed(7)
    mk = lambda: network_architectures.seg.get(arch)(C, pretrained=False).to(dev)
    stu, tea = mk(), mk()
    if os.environ.get('DBG_HIP'):
        stu.engine_kind = tea.engine_kind = 'hip'
    opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=1e-4), dict(params=list(stu.new_parameters()), lr=1e-3)])
    for p in tea.parameters(): p.requires_grad = False
    ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99); ema.fuse_into(opt)
    stu.train(); tea.train(); stu.freeze_batchnorm(); tea.freeze_batchnorm()
    step = CutMixMeanTeacherStep(stu, tea, opt, ema, StepConfig(conf_thresh=0.3, deterministic=True))
    return stu, opt, step
def batches(it):
    g = torch.Generator(device=dev).manual_seed(it)
    im = lambda: torch.randn(N, 3, H, W, generator=g, device=dev).bfloat16()
    y = torch.randint(0, C, (N, 1, H, W), generator=g, device=dev).to(torch.uint8)
    r = mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(N, (H, W), rng=np.random.RandomState(it))
    return im(), y, [UnsupBatch(im(), ops.ranges_to_device(r, dev), x1_tea=im())]
outs = []
for rep in range(4):
    stu, opt, step = build()
    cap = {}
    stu.hip_executor().debug_capture = cap
    torch.manual_seed(4242)
    res = step(*batches(0))
    torch.cuda.synchronize()
    sums = []
    ex = stu.hip_executor()
    for fp in ex.programs():
        for bw in getattr(fp, 'bwd', {}).values():
            for nm in ('dC_in', 'dlow_in'):
                t = getattr(bw, nm, None)
                if t is not None:
                    sums.append((nm, tuple(t.shape), float(t.double().abs().sum())))
    print('rep', rep, 'head->backbone gradients', sums)
    caps = {bi: tuple(float(t.double().abs().sum()) for t in v) for bi, v in cap.items()}
    wsum = {sg.key: float(opt.arena.view(sg.key, opt.arena.grad).double().abs().sum()) for sg in opt.arena.segments if sg.requires_grad}
    outs.append((opt.arena.grad.clone(), {k: float(v) for k, v in res.items()}, caps, wsum))
arena = opt.arena
for rep in range(1, 4):
    g0, g = outs[0][0], outs[rep][0]
    worst = sorted(((float((arena.view(sg.key, g) - arena.view(sg.key, g0)).abs().max() / g0.abs().max()), sg.key)
                    for sg in arena.segments if sg.requires_grad), reverse=True)[:6]
    nz = sum(1 for sg in arena.segments if sg.requires_grad and not torch.equal(arena.view(sg.key, g), arena.view(sg.key, g0)))
    print('   differing tensors', nz, worst)
    print('rep', rep, 'max rel grad diff', float((g - g0).abs().max() / g0.abs().max()), outs[rep][1], outs[0][1])

c0, c1 = outs[0][2], outs[1][2]
for bi in sorted(c0, reverse=True):
    if c0[bi] != c1[bi]:
        print('first differing block (backward order):', bi, c0[bi], c1[bi]); break
else:
    print('data-gradient chain identical in rep 0 and rep 1 for all', len(c0), 'blocks')
w0, w1 = outs[0][3], outs[1][3]
bad = [k for k in w0 if w0[k] != w1[k]]
print(len(bad), 'weight-gradient tensors differ; last few in state-dict order:', bad[-6:])
