import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, torch.nn.functional as F
from test_gpu_deeplab3plus import _he_state, _net
from oracle import deeplab3plus as o3
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from _library_engine import LibraryEngine as TorchEngine
DEV = 'cuda:0'
layers, C = (2, 2, 3, 2), 6
st = _he_state(C, layers)
g = torch.Generator().manual_seed(9)
x = torch.randn(3, 3, 97, 129, generator=g)
hip = _net(C, layers, torch.bfloat16, st); hip.eval()
# oracle per-block outputs
xx = F.conv2d(x, st[o3.B + 'conv1.weight'], stride=2, padding=3)
xx = F.relu(o3._bn(xx, st, o3.B + 'bn1', True, None))
xx = F.max_pool2d(xx, 3, 2, 1)
ref_blocks = []
cur = xx
for (pre, inpl, pl, stride, dil, down) in o3.layer_plan(layers):
    out = F.relu(o3._bn(F.conv2d(cur, st[pre + '.conv1.weight']), st, pre + '.bn1', True, None))
    out = F.conv2d(out, st[pre + '.conv2.weight'], stride=stride, padding=dil, dilation=dil)
    out = F.relu(o3._bn(out, st, pre + '.bn2', True, None))
    out = o3._bn(F.conv2d(out, st[pre + '.conv3.weight']), st, pre + '.bn3', True, None)
    res = cur
    if down:
        res = o3._bn(F.conv2d(cur, st[pre + '.downsample.0.weight'], stride=stride), st, pre + '.downsample.1', True, None)
    cur = F.relu(out + res)
    ref_blocks.append((pre, stride, dil, cur))
with torch.no_grad():
    eng = TorchEngine(torch.bfloat16)
    bb = hip.deeplab.backbone
    y = eng.conv_bn_act(eng.prepare_input(x.to(DEV)), bb['conv1'], bb['bn1'], relu=True)
    y = F.max_pool2d(y, 3, 2, 1)
    print('stem rel', float((y.float().cpu() - xx).norm() / xx.norm()))
    ex = hip.hip_executor()
    s = ex.fwd_begin(y.permute(0, 2, 3, 1).contiguous(), False)
    # library path block by block
    yl = y
    blocks_lib = [b for name in ('layer1', 'layer2', 'layer3', 'layer4') for b in bb[name]]
    for bi in range(len(ex.blocks)):
        ex.fwd_block(s, bi)
        yl = blocks_lib[bi](yl, eng)
        pre, stride, dil, r = ref_blocks[bi]
        h = s['cur'].permute(0, 3, 1, 2).float().cpu()
        l = yl.float().cpu()
        print(pre.split('backbone.')[1], 's', stride, 'd', dil, 'hip rel %.4f  lib rel %.4f  hip-vs-lib %.4f' % (
            float((h - r).norm() / r.norm()), float((l - r).norm() / r.norm()), float((h - l).norm() / l.norm())), tuple(h.shape))
