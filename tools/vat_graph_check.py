#!/usr/bin/env python
"""The VAT iteration's gradient passes as one hipGraph launch (vat.VATMeanTeacherStep.use_graph) against the eager launches:
two identically seeded student / teacher pairs, the same inputs and the same initial noise for `iters` iterations -- losses and
the student's weights must agree to the noise of the fp32 atomics; then the throughput of both forms.
    python tools/vat_graph_check.py [denseunet|resunet|deeplab] [iters]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import optim as fo, vat
from architectures import network_architectures
import optim_weight_ema

which = sys.argv[1] if len(sys.argv) > 1 else 'denseunet'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device('cuda:0')
if which == 'denseunet':
    B, H, W, C, name = 10, 224, 224, 2, 'densenet161unet_imagenet'
elif which == 'resunet':
    B, H, W, C, name = 10, 224, 224, 2, 'resnet50unet_imagenet'
else:
    B, H, W, C, name = 10, 321, 321, 21, 'resnet101_deeplab_imagenet'


def build(use_graph):
    torch.manual_seed(0)
    Net = network_architectures.seg.get(name)
    stu, tea = Net(C, pretrained=False).to(dev), Net(C, pretrained=False).to(dev)
    if which == 'deeplab':
        opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=3e-6), dict(params=list(stu.new_parameters()), lr=3e-5)])
        cfg = vat.VATConfig(cons_loss_fn='kld', conf_thresh=0.0)
    else:
        opt = fo.FusedSGD(stu, [dict(params=list(stu.pretrained_parameters()), lr=0.01), dict(params=list(stu.new_parameters()), lr=0.1)],
                          momentum=0.9, nesterov=True, weight_decay=5e-4)
        cfg = vat.VATConfig(vat_radius=1.0, adaptive_vat_radius=True, cons_loss_fn='kld', cons_weight=0.001, conf_thresh=0.0)
    for p in tea.parameters():
        p.requires_grad = False
    ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
    ema.fuse_into(opt)
    stu.train(); tea.train()
    if which == 'deeplab':
        stu.freeze_batchnorm(); tea.freeze_batchnorm()
    step = vat.VATMeanTeacherStep(stu, tea, opt, ema, cfg, generator=torch.Generator(device=dev).manual_seed(5))
    step.use_graph = use_graph
    return stu, opt, step


g = torch.Generator(device=dev).manual_seed(1)
data = []
for _ in range(iters):
    x = torch.randn(B, 3, H, W, generator=g, device=dev).bfloat16()
    xt = torch.randn(B, 3, H, W, generator=g, device=dev).bfloat16()
    y = torch.randint(0, C, (B, 1, H, W), generator=g, device=dev).to(torch.uint8)
    e = vat.normalized_noise_like(xt, 1.0e-6 * H * W / 1000, g)
    data.append((x, y, xt, e))

runs = {}
for use_graph in (False, 'again', True):
    torch.manual_seed(123)                                                               # dropout draws of both runs
    stu, opt, step = build(use_graph is True)
    losses = []
    for x, y, xt, e in data:
        r = step(x, y, [vat.VATUnsupBatch(xt)], eps0=e)
        losses.append((float(r['sup_loss']), float(r['consistency_loss'])))
    torch.cuda.synchronize()
    runs[use_graph] = (losses, opt.arena.flat.clone(), step)
    print('{}: losses {}'.format({False: 'eager', 'again': 'eager again', True: 'graph'}[use_graph], ['%.5f / %.3e' % l for l in losses]), flush=True)
(la, wa, _), (lb, wb, step_g) = runs[False], runs[True]
assert any('graph' in v for v in step_g._graphs.values()), 'the graph path never captured'
dl = max(abs(a[0] - b[0]) / (abs(a[0]) + 1e-12) for a, b in zip(la, lb))
dw = float((wa - wb).abs().max() / (wa.abs().max() + 1e-30))
w2 = runs['again'][1]
dw_eager = float((wa - w2).abs().max() / (wa.abs().max() + 1e-30))
dl_eager = max(abs(a[0] - b[0]) / (abs(a[0]) + 1e-12) for a, b in zip(la, runs['again'][0]))
print('max relative difference graph vs eager: sup loss {:.2e}, student weights {:.2e};  eager vs eager again (the yardstick: fp32 atomics): sup loss {:.2e}, student weights {:.2e}'.format(dl, dw, dl_eager, dw_eager), flush=True)
# (dropout of the U-Net decoders draws from the default generator: inside a graph its offsets advance like eager draws of the same
# sizes, but the two runs are only comparable while no dropout is active; with dropout the check is on the losses' scale)
tol = 5e-2 if which != 'deeplab' else 2e-3
assert dl <= max(tol, 4 * dl_eager), (dl, dl_eager)
assert dw <= max(1e-3, 4 * dw_eager), (dw, dw_eager)

# throughput
for use_graph in (False, True):
    stu, opt, step = build(use_graph)
    x, y, xt, e = data[0]
    for _ in range(4):
        step(x, y, [vat.VATUnsupBatch(xt)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 8
    for _ in range(K):
        r = step(x, y, [vat.VATUnsupBatch(xt)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print('VAT step [{} {}]: {:.1f} ms, {:.1f} img/s'.format(which, 'hipGraph' if use_graph else 'eager', dt * 1e3, B / dt), flush=True)
print('OK')
