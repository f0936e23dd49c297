#!/usr/bin/env python
"""VAT mean-teacher iteration throughput at BASELINE configs[1] geometry (DeepLab v2 / ResNet-101, 10 x 3 x 321 x 321):
the iteration in bf16 on the MFMA executor, the direction pass in fp32 on the library engine (vat.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import optim as fo, vat
from architectures import network_architectures
import optim_weight_ema

dev = torch.device('cuda:0')
B, H, W, C = 10, 321, 321, 21
torch.manual_seed(0)
Net = network_architectures.seg.get('resnet101_deeplab_imagenet')
stu, tea = Net(C, pretrained=False).to(dev), Net(C, pretrained=False).to(dev)
opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=3e-6), dict(params=list(stu.new_parameters()), lr=3e-5)])
for p in tea.parameters():
    p.requires_grad = False
ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
ema.fuse_into(opt)
stu.train(); tea.train(); stu.freeze_batchnorm(); tea.freeze_batchnorm()
g = torch.Generator(device=dev).manual_seed(1)
step = vat.VATMeanTeacherStep(stu, tea, opt, ema, vat.VATConfig(cons_loss_fn='kld', conf_thresh=0.97), generator=g)
im = lambda: torch.randn(B, 3, H, W, generator=g, device=dev).bfloat16()
y = torch.randint(0, C, (B, 1, H, W), generator=g, device=dev).to(torch.uint8)
x, xt = im(), im()
for _ in range(3):
    step(x, y, [vat.VATUnsupBatch(xt)])
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 8
for _ in range(K):
    r = step(x, y, [vat.VATUnsupBatch(xt)])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print('VAT step: {:.1f} ms, {:.1f} img/s (sup loss {:.3f})'.format(dt * 1e3, B / dt, float(r['sup_loss'])))
