import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from cutmix_semisup_seg_amd import ops
DEV='cuda:0'; N=20
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
KS=(0,12,16,20,24,28,32,40,48,56,64,80)
print('%-20s'%'shape'+''.join('%8s'%('ks=%d'%k) for k in KS))
for name,H,W,Cin,Cout,k,dil in [('l3 1x1 1024->256',41,41,1024,256,1,1),('l3 3x3d2 256->256',41,41,256,256,3,2),('l3 1x1 256->1024',41,41,256,1024,1,1),('l2 1x1 128->512',41,41,128,512,1,1),('l4 1x1 2048->512',41,41,2048,512,1,1),('c3 l3 1x1 1024->256',65,129,1024,256,1,1)]:
    n = 8 if name.startswith('c3') else N
    g=torch.Generator(device=DEV).manual_seed(0); pad=dil*(k-1)//2
    x=torch.randn(n,H,W,Cin,generator=g,device=DEV).bfloat16(); du=torch.randn(n,H,W,Cout,generator=g,device=DEV).bfloat16()
    dw=torch.zeros(k*k,Cout,Cin,device=DEV); taps=ops.conv_taps(k,k,dil,pad)
    ts=[]
    for ks in KS:
        try: ts.append(timeit(lambda: ops.conv_wgrad(du,x,taps,dw,ksplit=ks)))
        except Exception as e: ts.append(float('nan'))
    print('%-20s'%name+''.join('%8.1f'%t for t in ts))
