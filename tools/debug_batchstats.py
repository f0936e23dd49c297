"""Debug: per-layer gradient error of DeepLab v2 on batch statistics, engines 'torch' (library convs + bn.hip) and 'hip'
(hand-written convs + bn.hip), fp32, against the CPU oracle's autograd -- one supervised pass."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np, torch
from oracle import deeplab2 as dl, losses as L
from architectures import deeplab2

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
y = torch.randint(0, C, (N, 1, H, W), generator=g)
keys = dl.trainable_keys(C, layers)
leaves = {k: st[k].clone().requires_grad_(True) for k in keys}
s2 = dict(st); s2.update(leaves)
out = dl.forward_lowres(x, s2, layers, frozen=False, new_stats={})
tgt = torch.randn(out.shape, generator=g)
((out - tgt) ** 2).mean().backward()
for kind in ('torch', 'hip'):
    net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    net.load_state_dict(st); net = net.cuda(); net.compute_dtype = torch.float32; net.engine_kind = kind; net.train()
    for p in net.parameters():
        if p.requires_grad: p.grad = None
    lo = net.forward_lowres(x.cuda())
    ((lo - tgt.cuda()) ** 2).mean().backward()
    print(kind, 'logits rel %.2e' % float((lo.detach().cpu() - out.detach()).norm() / out.detach().norm()))
    for k, p in net.named_parameters():
        if p.grad is not None and leaves.get(k) is not None and leaves[k].grad is not None:
            gd, gr = p.grad.cpu().double().flatten(), leaves[k].grad.double().flatten()
            sc = float((gd @ gr) / (gr @ gr))
            print('   %-34s %.2e   scale-1 %+.2e  residual after rescale %.2e' % (
                k, float((gd - gr).norm() / gr.norm()), sc - 1.0, float((gd - sc * gr).norm() / gr.norm())))

# the oracle's own functional code on the GPU (library fp32 kernels end to end): how far does plain PyTorch-GPU land from
# PyTorch-CPU on this problem?
leaves_g = {k: st[k].clone().cuda().requires_grad_(True) for k in keys}
s3 = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in st.items()}
s3.update(leaves_g)
out_g = dl.forward_lowres(x.cuda(), s3, layers, frozen=False, new_stats={})
((out_g - tgt.cuda()) ** 2).mean().backward()
print('aten-gpu logits rel %.2e' % float((out_g.detach().cpu() - out.detach()).norm() / out.detach().norm()))
for k in keys:
    if leaves[k].grad is not None:
        print('   %-34s %.2e' % (k, float((leaves_g[k].grad.cpu() - leaves[k].grad).norm() / leaves[k].grad.norm())))
