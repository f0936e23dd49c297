import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from test_gpu_executor import _build, _cf_input, _rel
DEV = 'cuda:0'
layers, C = [1, 1, 1, 1], 5
active = sys.argv[1] == 'linear' if len(sys.argv) > 1 else True
x = _cf_input(2, 65, 81, 0.7).bfloat16().to(DEV)
hip, lib, ref = _build(layers, C, 'hip', active), _build(layers, C, 'torch', active), _build(layers, C, 'torch', active)
ref.compute_dtype = torch.float32


def run_lib(net):
    outs, grads = [], {}
    hooks = []
    def mk(i):
        def hook(mod, inp, out):
            out.register_hook(lambda g, i=i: grads.__setitem__(i, g.detach().float().permute(0, 2, 3, 1)))
        return hook
    i = 0
    for li in range(1, 5):
        for blk in getattr(net, 'layer{}'.format(li)):
            hooks.append(blk.register_forward_hook(mk(i)))
            i += 1
    lo = net.forward_lowres(x)
    return lo, grads

lo_l, g_l = run_lib(lib)
lo_r, g_r = run_lib(ref)
lo_h = hip.forward_lowres(x)
ex = hip._hip_executor
ex.debug_capture = {}
g = torch.randn(lo_h.shape, generator=torch.Generator(device=DEV).manual_seed(1), device=DEV)
hip._cms_arena.zero_grad()
lo_h.backward(g); lo_l.backward(g); lo_r.backward(g)
print('forward rel err vs fp32: hip', _rel(lo_h, lo_r), 'lib', _rel(lo_l, lo_r))
for bi in sorted(g_r.keys(), reverse=True):
    print('block', bi, 'grad wrt block output: hip', round(_rel(ex.debug_capture[bi].float(), g_r[bi]), 4), 'lib', round(_rel(g_l[bi], g_r[bi]), 4),
          '| norms', float(g_r[bi].norm()), float(ex.debug_capture[bi].float().norm()))
nh, nl, nr = dict(hip.named_parameters()), dict(lib.named_parameters()), dict(ref.named_parameters())
for k in nr:
    if nr[k].grad is not None and nr[k].dim() == 4:
        print('{:<34s} hip {:.4f} lib {:.4f}'.format(k, _rel(nh[k].grad, nr[k].grad), _rel(nl[k].grad.float(), nr[k].grad)))
