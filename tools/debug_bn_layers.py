"""Debug: every batch-statistics BatchNorm call of one DeepLab v2 pass (engine 'hip', fp32) checked on its own against the
fp64 formula on the SAME inputs / incoming gradient (hooks) -- isolates csrc/bn.hip per layer on real activations."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
from oracle import deeplab2 as dl
from architectures import deeplab2
from cutmix_semisup_seg_amd import ops

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
orig = ops.batch_norm_act


def wrapped(xh, gamma, beta, rm, rv, momentum=0.1, eps=1e-5, relu=False, res=None, group=None):
    y = orig(xh, gamma, beta, rm, rv, momentum, eps, relu, res, group)
    r = dict(x=xh.detach().clone(), res=None if res is None else res.detach().clone(), y=y.detach().clone(),
             gamma=gamma.detach().clone(), beta=beta.detach().clone(), relu=relu, eps=eps, C=xh.shape[-1])
    if xh.requires_grad:
        xh.register_hook(lambda gr, r=r: r.__setitem__('dx', gr.detach().clone()))
    y.register_hook(lambda gr, r=r: r.__setitem__('dy', gr.detach().clone()))
    recs.append(r)
    return y


ops.batch_norm_act = wrapped
net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
net.load_state_dict(st); net = net.cuda(); net.compute_dtype = torch.float32; net.engine_kind = 'hip'; net.train()
lo = net.forward_lowres(x.cuda())
tgt = torch.randn(lo.shape, generator=g).cuda()
((lo - tgt) ** 2).mean().backward()
rel = lambda a, b: float((a.double() - b).norm() / (b.norm() + 1e-300))
for i, r in enumerate(recs):
    xd = r['x'].double().cpu(); C_ = r['C']
    flat = xd.reshape(-1, C_)
    mean, var = flat.mean(0), flat.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + r['eps'])
    xhat = (xd - mean) * rstd
    yref = xhat * r['gamma'].double().cpu() + r['beta'].double().cpu()
    if r['res'] is not None: yref = yref + r['res'].double().cpu()
    mask = (yref > 0) if r['relu'] else torch.ones_like(yref, dtype=torch.bool)
    if r['relu']: yref = yref.clamp_min(0)
    line = 'bn %2d C=%4d relu=%d res=%d  y rel %.1e' % (i, C_, r['relu'], r['res'] is not None, rel(r['y'].cpu(), yref))
    if 'dy' in r and 'dx' in r:
        dy = r['dy'].double().cpu()
        # the device's own mask (its stored y), so that a flipped tie is not counted against the arithmetic
        dmask = (r['y'].cpu() > 0) if r['relu'] else mask
        d = dy * dmask
        m1, m2 = d.reshape(-1, C_).mean(0), (d * xhat).reshape(-1, C_).mean(0)
        dxref = r['gamma'].double().cpu() * rstd * (d - m1 - xhat * m2)
        line += '  dx rel %.1e  (|dx|/|dy| %.2e, min var %.2e, mask flips %d)' % (
            rel(r['dx'].cpu(), dxref), float(dxref.norm() / dy.norm()), float(var.min()), int((dmask != mask).sum()))
    print(line)
