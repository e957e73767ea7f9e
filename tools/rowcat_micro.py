import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd.ctx_ops import rowcat
from contextgs_amd import _lib
import ctypes as C
L=_lib.lib()
def prof(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); L.cgs_prof_enable(1)
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    out={}
    for i in range(L.cgs_prof_count()):
        ms,c=C.c_double(),C.c_int64(); L.cgs_prof_read(i,C.byref(ms),C.byref(c))
        if c.value: out[L.cgs_prof_name(i).decode()]=ms.value/c.value*1e3
    L.cgs_prof_enable(0); return out
n, p = 800_000, 200_000
a=torch.randn(1_000_000,3,device='cuda',requires_grad=True); f=torch.randn(p,50,device='cuda',requires_grad=True)
s=torch.randn(p,6,device='cuda',requires_grad=True); h=torch.randn(n,12,device='cuda',requires_grad=True)
idx=torch.randint(0,1_000_000,(n,),device='cuda'); pos=torch.randint(0,p,(n,),device='cuda').sort()[0]
g=torch.randn(n,71,device='cuda')
def step():
    y=rowcat([(a,idx,False),(f,pos,False),(s,pos,False),(h,None,True)])
    torch.autograd.grad(y,[a,f,s,h],g)
print("ctx4", prof(step), "MB", n*71*4*2/1e6)
orig=torch.randperm(1_000_000,device='cuda')[:n]
def step2():
    y=rowcat([(a,orig,True),(h,None,True)])
    torch.autograd.grad(y,[a,h],g[:,:15].contiguous())
print("ctx2", prof(step2))
