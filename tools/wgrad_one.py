import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cutmix_semisup_seg_amd import ops
H, W, Cin, Cout, k, dil = (int(v) for v in sys.argv[1:7])
ks = int(sys.argv[7]) if len(sys.argv) > 7 else 0
DEV = 'cuda:0'; N = 20
g = torch.Generator(device=DEV).manual_seed(0)
pad = dil * (k - 1) // 2
x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
du = torch.randn(N, H, W, Cout, generator=g, device=DEV).bfloat16()
dw = torch.zeros(k * k, Cout, Cin, device=DEV)
for _ in range(5):
    ops.conv_wgrad(du, x, ops.conv_taps(k, k, dil, pad), dw, ksplit=ks)
torch.cuda.synchronize()
