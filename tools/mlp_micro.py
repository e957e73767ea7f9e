"""Per-shape timing of the fused MLP kernels (forward / backward / weight gradients) with the
library's own HIP-event scopes.  python tools/mlp_micro.py [n]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd import _lib, mlp

L = _lib.lib()
names = [L.cgs_prof_name(i).decode() for i in range(L.cgs_prof_count())]

def read():
    out = {}
    for i, nm in enumerate(names):
        ms, cnt = C.c_double(), C.c_int64()
        L.cgs_prof_read(i, C.byref(ms), C.byref(cnt))
        if cnt.value:
            out[nm] = (ms.value, cnt.value)
    return out

def run(cfg, n, reps=5):
    i, h, o, act = cfg
    dev = "cuda"
    x = torch.randn(n, i, device=dev, requires_grad=True)
    W1 = torch.randn(h, i, device=dev, requires_grad=True); b1 = torch.randn(h, device=dev, requires_grad=True)
    W2 = torch.randn(o, h, device=dev, requires_grad=True); b2 = torch.randn(o, device=dev, requires_grad=True)
    g = torch.randn(n, o, device=dev)
    for r in range(reps + 2):
        if r == 2:
            torch.cuda.synchronize(); L.cgs_prof_enable(1)
        y = mlp.mlp2_weights(x, W1, b1, W2, b2, act)
        y.backward(g)
    torch.cuda.synchronize()
    p = read(); L.cgs_prof_enable(0)
    bytes_f = n * 4 * (i + h + o); bytes_b = n * 4 * (o * (3 if act else 1) + 2 * h + i); bytes_w = n * 4 * (o + h + h + i)
    fl = 2 * n * (i * h + h * o)
    f = p["mlp_fwd"][0] / reps * 1e3; b = p["mlp_bwd"][0] / reps * 1e3; w = p["mlp_wgrad"][0] / reps * 1e3
    print(f"{cfg} n={n}: fwd {f:7.1f} us ({bytes_f/f/1e6:5.2f} TB/s, {fl/f/1e6:5.1f} TF)  bwd {b:7.1f} us ({bytes_b/b/1e6:5.2f} TB/s, {2*fl/b/1e6:5.1f} TF)"
          f"  wgrad {w:7.1f} us ({bytes_w/w/1e6:5.2f} TB/s, {fl/w/1e6:5.1f} TF)")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600_000
for cfg in [(54, 50, 10, 1), (54, 50, 30, 2), (54, 50, 70, 0), (71, 100, 3, 0), (15, 100, 3, 0), (71, 100, 175, 0), (15, 100, 175, 0)]:
    run(cfg, n)
run((71, 100, 175, 0), 60_000)
run((71, 100, 3, 0), 100_000)
