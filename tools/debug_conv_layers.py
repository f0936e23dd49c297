"""Debug: every convolution call of one DeepLab v2 pass (engine 'hip', fp32, batch statistics) checked on its own against
torch's fp64 convolution on the SAME input / incoming gradient (hooks): forward, data gradient, weight gradient."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import torch.nn.functional as F
from oracle import deeplab2 as dl
from architectures import deeplab2
from cutmix_semisup_seg_amd import backbone_hip

C, layers, N, H, W = 5, [1, 1, 1, 1], 3, 49, 65
g = torch.Generator().manual_seed(77)
st = {}
for k, (shape, dt) in dl.state_spec(C, layers).items():
    if dt == torch.int64: st[k] = torch.zeros(shape, dtype=torch.int64)
    elif len(shape) == 4: st[k] = torch.randn(shape, generator=g) * (2.0 / (shape[1] * shape[2] * shape[3])) ** 0.5 * (0.3 if k.startswith('layer5.') else 1.0)
    elif k.endswith('running_var'): st[k] = 0.8 + 0.4 * torch.rand(shape, generator=g)
    elif k.endswith('running_mean'): st[k] = 0.1 * torch.randn(shape, generator=g)
    elif k.endswith('.weight'): st[k] = 0.6 + 0.8 * torch.rand(shape, generator=g)
    else: st[k] = 0.1 * torch.randn(shape, generator=g)
g = torch.Generator().manual_seed(21)
x = torch.randn(N, 3, H, W, generator=g)
recs = []
orig = backbone_hip.hip_conv2d


def wrapped(xin, conv, arena, key, dtype=torch.bfloat16):
    y = orig(xin, conv, arena, key, dtype)
    r = dict(key=key, x=xin.detach().clone(), y=y.detach().clone(), conv=conv)
    if xin.requires_grad:
        xin.register_hook(lambda gr, r=r: r.__setitem__('dx_total', gr.detach().clone()))
    y.register_hook(lambda gr, r=r: r.__setitem__('dy', gr.detach().clone()))
    recs.append(r)
    return y


backbone_hip.hip_conv2d = wrapped
net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
net.load_state_dict(st); net = net.cuda(); net.compute_dtype = torch.float32; net.engine_kind = 'hip'; net.train()
lo = net.forward_lowres(x.cuda())
tgt = torch.randn(lo.shape, generator=g).cuda()
((lo - tgt) ** 2).mean().backward()
rel = lambda a, b: float((a.double().cpu() - b).norm() / (b.norm() + 1e-300))
params = dict(net.named_parameters())
for r in recs:
    conv = r['conv']
    xd = r['x'].double().cpu().requires_grad_(True)
    wd = params[r['key']].detach().double().cpu().requires_grad_(True)
    yref = F.conv2d(xd, wd, None, conv.stride, conv.padding, conv.dilation)
    line = '%-30s k%d s%d d%d %4d->%4d %2dx%2d  y %.1e' % (r['key'], conv.kernel_size[0], conv.stride[0], conv.dilation[0],
                                                        conv.in_channels, conv.out_channels, r['x'].shape[2], r['x'].shape[3],
                                                        rel(r['y'], yref.detach()))
    if 'dy' in r:
        yref.backward(r['dy'].double().cpu())
        line += '  dW %.1e' % rel(params[r['key']].grad, wd.grad)
        line += '  |dx_conv| %.2e' % float(xd.grad.norm())
        if 'dx_total' in r and ('.conv2.' in r['key'] or '.conv3.' in r['key']):      # inputs used by this convolution only
            line += '  dx %.1e' % rel(r['dx_total'], xd.grad)
    print(line)
