#!/usr/bin/env python
"""CutMix mean-teacher iteration (train_seg_semisup_mask_mt.py:287-476) of a U-Net through the layer engines: throughput and how
much of the step the host needs to enqueue it.
    python tools/unet_cutmix_bench.py [resnet50unet_imagenet|densenet161unet_imagenet] [B] [H]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cutmix_semisup_seg_amd import ops, optim as fo
from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig, UnsupBatch
from architectures import network_architectures
import mask_gen
import optim_weight_ema

name = sys.argv[1] if len(sys.argv) > 1 else 'resnet50unet_imagenet'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 10
H = W = int(sys.argv[3]) if len(sys.argv) > 3 else 256
C = 2
dev = torch.device('cuda:0')
torch.manual_seed(0)
Net = network_architectures.seg.get(name)
stu, tea = Net(C, pretrained=False).to(dev), Net(C, pretrained=False).to(dev)
opt = fo.FusedSGD(stu, [dict(params=list(stu.pretrained_parameters()), lr=0.01), dict(params=list(stu.new_parameters()), lr=0.1)],
                  momentum=0.9, nesterov=True, weight_decay=5e-4)
for p in tea.parameters():
    p.requires_grad = False
ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
ema.fuse_into(opt)
stu.train(); tea.train()
step = CutMixMeanTeacherStep(stu, tea, opt, ema, StepConfig(conf_thresh=0.97))
g = torch.Generator(device=dev).manual_seed(1)
im = lambda: torch.randn(B, 3, H, W, generator=g, device=dev).bfloat16()
y = torch.randint(0, C, (B, 1, H, W), generator=g, device=dev).to(torch.uint8)
x, x0, x1 = im(), im(), im()
rng = np.random.RandomState(3)


def one():
    r = mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(B, (H, W), rng=rng)
    return step(x, y, [UnsupBatch(x0, ops.ranges_to_device(r, dev), x1_tea=x1)])


for _ in range(int(os.environ.get('WARMUP', '5'))):
    one()
torch.cuda.synchronize()
K = int(os.environ.get('STEPS', '12'))
t0 = time.perf_counter()
for _ in range(K):
    res = one()
t_host = (time.perf_counter() - t0) / K
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print('CutMix step [{} {}x3x{}x{}]: {:.1f} ms per step, {:.1f} img/s; host enqueue {:.1f} ms per step (sup loss {:.3f})'.format(
    name, B, H, W, dt * 1e3, B / dt, t_host * 1e3, float(res['sup_loss'])))
