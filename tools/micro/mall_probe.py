"""Does a buffer written by one kernel and read by the next stay in the 256 MB Infinity Cache (MALL)?  Times, for buffer
sizes from 16 MB to 2 GB, a write pass (fill), a read pass (sum) right after it, and a copy, reusing the SAME buffer
(the pattern a row-chunked producer/consumer pipeline with a recycled scratch buffer would have).  GPU box."""
import torch

dev = "cuda"
def t(fn, reps):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3

print(f"{'MB':>6} {'write GB/s':>11} {'read-after-write GB/s':>22} {'copy (r+w) GB/s':>16} {'pair: fill+sum GB/s (2x bytes)':>32}")
for mb in (16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    reps = max(5, 4096 // mb)
    a.fill_(1.0); a.sum()
    tw = t(lambda: a.fill_(1.0), reps)
    tr = t(lambda: a.sum(), reps)
    tc = t(lambda: b.copy_(a), reps)
    def pair():
        a.fill_(2.0)
        a.sum()
    tp = t(pair, reps)
    gb = n * 4 / 1e9
    print(f"{mb:6d} {gb / tw:11.0f} {gb / tr:22.0f} {2 * gb / tc:16.0f} {2 * gb / tp:32.0f}")
    del a, b
