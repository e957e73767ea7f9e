import torch, time, sys
sys.path.insert(0,'/root/repo')
from contextgs_amd import _lib
L=_lib.lib()
N=1_000_000
m=torch.rand(N,device='cuda')<0.995
idx=torch.nonzero(m)[:,0]
n=idx.numel()
for w in (3,10,30,50):
    g=torch.randn(n,w,device='cuda'); out=torch.empty(N,w,device='cuda')
    def a():
        o=torch.zeros(N,w,device='cuda'); o.index_copy_(0,idx,g); return o
    def b():
        _lib.check(L.cgs_scatter_rows_sorted(_lib.ptr(g),_lib.ptr(idx),n,N,w,_lib.ptr(out),_lib.current_stream()),"x"); return out
    assert torch.equal(a(),b())
    for f,name in ((a,'zeros+index_copy'),(b,'sorted scatter')):
        for _ in range(3): f()
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize(); print(w,name,(time.perf_counter()-t)/20*1e6,'us')
