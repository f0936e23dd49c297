import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, torch.nn.functional as F
from collections import OrderedDict
from test_gpu_deeplab3plus import _he_state, _net
from oracle import deeplab3plus as o3
layers, C = (1, 1, 1, 1), 4
st = _he_state(C, layers)
g = torch.Generator().manual_seed(11)
x = torch.randn(3, 3, 49, 65, generator=g)
y = torch.randint(0, C, (3, 49, 65), generator=g)
keys = o3.trainable_keys(C, layers)
leaves = {k: st[k].clone().requires_grad_(True) for k in keys}
s2 = OrderedDict(st); s2.update(leaves)
for mode in ('eval', 'train'):
    for k in leaves: leaves[k].grad = None
    out = o3.forward(x, s2, layers, True, mode == 'eval', {})
    F.cross_entropy(out, y).backward()
    ref = {k: leaves[k].grad.clone() for k in keys}
    for tf32 in (True, False):
        torch.backends.cudnn.allow_tf32 = tf32
        net = _net(C, layers, torch.float32, st)
        net.train() if mode == 'train' else net.eval()
        if mode == 'train': net.freeze_batchnorm()
        lo = net.forward_lowres(x.cuda())
        out_g = F.interpolate(lo, size=(49, 65), mode='bilinear', align_corners=False)
        F.cross_entropy(out_g, y.cuda()).backward()
        rels = []
        for k, p in net.named_parameters():
            rels.append(float((p.grad.cpu() - ref[k]).norm() / ref[k].norm()))
        import numpy as np
        print(mode, 'tf32', tf32, 'fwd rel', float((out_g.detach().cpu() - out.detach()).norm() / out.detach().norm()),
              'grad rel median', np.median(rels), 'max', max(rels), 'first', rels[0], 'last', rels[-2])
