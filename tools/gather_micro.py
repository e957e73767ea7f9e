import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd.ctx_ops import rowcat
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e6
N=1_000_000
for W in (50, 6, 30, 12, 3):
    x=torch.randn(N,W,device='cuda'); perm=torch.randperm(N,device='cuda'); sub=perm[:600_000].sort()[0]
    g=torch.randn(N,W,device='cuda')
    for name,idx in (('perm',perm),('sorted600k',sub)):
        gi=g[:idx.shape[0]].contiguous()
        a=t(lambda: x.index_select(0,idx)); b=t(lambda: rowcat([(x,idx,True)]))
        def tb():
            o=torch.zeros(N,W,device='cuda'); o.index_copy_(0,idx,gi); return o
        xr=x.clone().requires_grad_()
        y=rowcat([(xr,idx,True)])
        c=t(tb); d=t(lambda: torch.autograd.grad(y,[xr],gi,retain_graph=True))
        mb=idx.shape[0]*W*4*2/1e6
        print(f"W={W:3d} {name:10s} fwd torch {a:7.1f} us  rowcat {b:7.1f} us | bwd torch {c:7.1f}  rowcat {d:7.1f}   ({mb:.0f} MB moved)")
