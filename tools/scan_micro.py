"""cgs_scan_exclusive_u32: the single-pass (decoupled look-back) kernel against reduce + apply, by size.
CGS_SCAN_CHAINED_MIN_TILES=0 forces the single pass, =999999999 the two-launch path (read at first use: one process each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd import _lib
L = _lib.lib()
for n in (362_000, 1_000_000, 5_800_000, 10_000_000, 23_000_000):
    x = torch.randint(0, 50, (n,), device="cuda", dtype=torch.int32)
    out = torch.empty_like(x)
    scratch = torch.empty(L.cgs_scan_scratch_bytes(n), dtype=torch.uint8, device="cuda")
    f = lambda: _lib.check(L.cgs_scan_exclusive_u32(_lib.ptr(x), _lib.ptr(out), n, _lib.ptr(scratch), scratch.numel(), _lib.current_stream()), "scan")
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    ok = torch.equal(out.to(torch.int64), torch.cumsum(x.to(torch.int64), 0) - x)
    print(f"n {n:9d}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per scan  exact {ok}")
