#!/usr/bin/env python
"""VAT mean-teacher iteration throughput (train_seg_semisup_vat_mt.py iteration):
    python tools/vat_bench.py deeplab     DeepLab v2 / ResNet-101 at BASELINE configs[1] geometry (10 x 3 x 321 x 321): bf16
                                          iteration on the MFMA executor, direction pass on the fp32 MFMA executor
    python tools/vat_bench.py denseunet   BASELINE configs[4]: DenseNet-161 U-Net, 10 x 3 x 224 x 224, 2 classes, SGD
                                          (run_isic2017_experiments.sh:14-18), batch-statistics BatchNorm on csrc/bn.hip"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import optim as fo, vat
from architectures import network_architectures
import optim_weight_ema

which = sys.argv[1] if len(sys.argv) > 1 else 'deeplab'
engine_kind = sys.argv[2] if len(sys.argv) > 2 else None      # 'hip': every convolution on the hand-written kernels
dev = torch.device('cuda:0')
torch.manual_seed(0)
if which == 'denseunet':
    B, H, W, C = 10, 224, 224, 2
    Net = network_architectures.seg.get('densenet161unet_imagenet')
    stu, tea = Net(C, pretrained=False).to(dev), Net(C, pretrained=False).to(dev)
    opt = fo.FusedSGD(stu, [dict(params=list(stu.pretrained_parameters()), lr=0.01), dict(params=list(stu.new_parameters()), lr=0.1)],
                      momentum=0.9, nesterov=True, weight_decay=5e-4)
    cfg = vat.VATConfig(vat_radius=1.0, adaptive_vat_radius=True, cons_loss_fn='kld', cons_weight=0.001, conf_thresh=0.97)
else:
    B, H, W, C = 10, 321, 321, 21
    Net = network_architectures.seg.get('resnet101_deeplab_imagenet')
    stu, tea = Net(C, pretrained=False).to(dev), Net(C, pretrained=False).to(dev)
    opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=3e-6), dict(params=list(stu.new_parameters()), lr=3e-5)])
    cfg = vat.VATConfig(cons_loss_fn='kld', conf_thresh=0.97)
if engine_kind:
    stu.engine_kind = tea.engine_kind = engine_kind
for p in tea.parameters():
    p.requires_grad = False
ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
ema.fuse_into(opt)
stu.train(); tea.train()
if which != 'denseunet':
    stu.freeze_batchnorm(); tea.freeze_batchnorm()
g = torch.Generator(device=dev).manual_seed(1)
step = vat.VATMeanTeacherStep(stu, tea, opt, ema, cfg, generator=g)
im = lambda: torch.randn(B, 3, H, W, generator=g, device=dev).bfloat16()
y = torch.randint(0, C, (B, 1, H, W), generator=g, device=dev).to(torch.uint8)
x, xt = im(), im()
for _ in range(int(os.environ.get('VAT_BENCH_WARMUP', '5'))):     # (2 eager iterations + the hipGraph capture + replays)
    step(x, y, [vat.VATUnsupBatch(xt)])
torch.cuda.synchronize()
t0 = time.perf_counter()
K = int(os.environ.get('VAT_BENCH_STEPS', '16'))
for _ in range(K):
    r = step(x, y, [vat.VATUnsupBatch(xt)])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print('VAT step [{}{}]: {:.1f} ms, {:.1f} img/s (sup loss {:.3f}, cons loss {:.5f})'.format(
    which, ' engine_kind=' + engine_kind if engine_kind else '', dt * 1e3, B / dt, float(r['sup_loss']), float(r['consistency_loss'])))
if os.environ.get('CMS_HOST_PROFILE'):         # where the host time of an iteration goes
    import cProfile, pstats
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(x, y, [vat.VATUnsupBatch(xt)])
        ts.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    print('host ms per iteration on an empty queue:', ['%.1f' % t for t in ts])
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2):
        step(x, y, [vat.VATUnsupBatch(xt)])
        torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr, stream=sys.stdout).sort_stats('tottime').print_stats(25)
