"""Debug: gradient wrt every convolution OUTPUT (du) and every convolution output itself, device ('hip' fp32, batch statistics)
vs the CPU oracle, layer by layer -- finds the first tensor of the backward chain that deviates."""
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
# ---- oracle with hooks on every convolution output
keys = dl.trainable_keys(C, layers)
leaves = {k: st[k].clone().requires_grad_(True) for k in keys}
s2 = dict(st); s2.update(leaves)
name_of = {id(v): k for k, v in leaves.items()}
ref = {}
orig_conv = F.conv2d


def conv_rec(inp, w, *a, **k):
    y = orig_conv(inp, w, *a, **k)
    nm = name_of.get(id(w))
    if nm is not None:
        r = ref.setdefault(nm, {})
        r['y'] = y.detach().clone()
        r['x'] = inp.detach().clone()
        y.register_hook(lambda gr, r=r: r.__setitem__('du', gr.detach().clone()))
        if inp.requires_grad:
            inp.register_hook(lambda gr, r=r: r.__setitem__('dxin', gr.detach().clone()))
    return y


F.conv2d = conv_rec
dl.F.conv2d = conv_rec
out = dl.forward_lowres(x, s2, layers, frozen=False, new_stats={})
tgt = torch.randn(out.shape, generator=g)
((out - tgt) ** 2).mean().backward()
F.conv2d = orig_conv
dl.F.conv2d = orig_conv
# ---- device with hooks
dev = {}
orig = backbone_hip.hip_conv2d


def wrapped(xin, conv, arena, key, dtype=torch.bfloat16):
    y = orig(xin, conv, arena, key, dtype)
    r = dev.setdefault(key, {})
    r['y'] = y.detach().clone()
    r['x'] = xin.detach().clone()
    y.register_hook(lambda gr, r=r: r.__setitem__('du', gr.detach().clone()))
    if xin.requires_grad:
        xin.register_hook(lambda gr, r=r: r.__setitem__('dxin', gr.detach().clone()))
    return y


backbone_hip.hip_conv2d = wrapped
net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
net.load_state_dict(st); net = net.cuda(); net.compute_dtype = torch.float32; net.engine_kind = 'hip'; net.train()
lo = net.forward_lowres(x.cuda())
((lo - tgt.cuda()) ** 2).mean().backward()
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-300))
for k in keys:
    if k in dev and k in ref and 'du' in ref[k]:
        extra = ''
        if 'dxin' in dev[k] and 'dxin' in ref[k]:
            extra = '   x %.1e   d(input, all consumers) %.1e   sign(x) mismatches %d' % (
                rel(dev[k]['x'], ref[k]['x']), rel(dev[k]['dxin'], ref[k]['dxin']),
                int(((dev[k]['x'].cpu() > 0) != (ref[k]['x'] > 0)).sum()))
        print('%-32s y %.1e   du %.1e%s' % (k, rel(dev[k]['y'], ref[k]['y']), rel(dev[k]['du'], ref[k]['du']), extra))
