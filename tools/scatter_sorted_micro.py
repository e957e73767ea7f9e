"""cgs_scatter_rows_sorted at the step's shapes (1 M anchors, 99.5 % visible; w = 3 anchor rows, w = 10 mask rows).
CGS_LIB_PATH selects the library build.  python tools/scatter_sorted_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd import _lib

N = 1_000_000
torch.manual_seed(0)
idx = torch.nonzero(torch.rand(N, device="cuda") < 0.995)[:, 0]
n = int(idx.numel())
for w in (3, 10, 12):
    g = torch.randn(n, w, device="cuda"); out = torch.empty(N, w, device="cuda")
    f = lambda: _lib.check(_lib.lib().cgs_scatter_rows_sorted(_lib.ptr(g), _lib.ptr(idx), n, N, w, _lib.ptr(out), _lib.current_stream()), "s")
    for _ in range(5): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): f()
    b.record(); torch.cuda.synchronize()
    print("lib=%s w=%d  %.1f us" % (os.path.basename(os.environ.get("CGS_LIB_PATH", "product")), w, a.elapsed_time(b) / 50 * 1e3))
