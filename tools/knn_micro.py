"""Timing of the anchor-initialisation kNN (cgs_knn_mean_dist2) on uniform / clustered clouds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from contextgs_amd.knn import distCUDA2
for kind, n in (("uniform", 1_000_000), ("uniform", 5_000_000), ("clusters", 1_000_000)):
    rng = np.random.default_rng(0)
    if kind == "uniform":
        p = rng.random((n, 3), dtype=np.float32)
    else:
        c = rng.random((200, 3)) * 100
        p = (c[rng.integers(0, 200, n)] + rng.normal(0, 0.05, (n, 3)) * rng.random((n, 1)) ** 3).astype(np.float32)
    t = torch.from_numpy(p).cuda()
    distCUDA2(t); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        d = distCUDA2(t)
    torch.cuda.synchronize()
    print(f"{kind:9s} n={n:9d}  {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms   mean d2 {float(d.mean()):.3e}")
