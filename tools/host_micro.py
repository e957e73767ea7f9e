"""Host-side cost of the per-launch helpers (us per call): the stream lookup, pointers, allocations, one ctypes call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd import _lib

def t(fn, n=20000):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6

x = torch.empty(1000, 50, device="cuda")
L = _lib.lib()
print("torch.cuda.current_stream().cuda_stream  %.2f us" % t(lambda: torch.cuda.current_stream().cuda_stream))
print("_cuda_getCurrentRawStream(current_device) %.2f us" % t(lambda: torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())))
print("_lib.current_stream()                    %.2f us" % t(_lib.current_stream))
print("_lib.ptr(x)                              %.2f us" % t(lambda: _lib.ptr(x)))
print("_lib.lib()                               %.2f us" % t(_lib.lib))
print("_lib.require_device(x, x, x)             %.2f us" % t(lambda: _lib.require_device(x, x, x)))
print("torch.empty(1000, 50, cuda)              %.2f us" % t(lambda: torch.empty(1000, 50, dtype=torch.float32, device=x.device)))
print("x.contiguous()                           %.2f us" % t(lambda: x.contiguous()))
print("cgs_mlp_wgrad_scratch_bytes() (ctypes)   %.2f us" % t(lambda: L.cgs_mlp_wgrad_scratch_bytes()))
