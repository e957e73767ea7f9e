"""Micro-benchmark of the fused level kernels (csrc/ctx_level.hip) on level-0-sized operands: cgs_ctx_level_fwd / _bwd at
n rows of in_dim 71 (and the round-4 launches they replace, for the same rows).  python tools/ctxl_micro.py [n] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 800_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
in_dim = int(sys.argv[3]) if len(sys.argv) > 3 else 71
L = _lib.lib()
dev = "cuda"
p = _lib.ptr
g = torch.Generator(device="cpu").manual_seed(1)
N, n_par = n + n // 4, n // 4
R = lambda *s: torch.randn(*s, device=dev)
anchor, hyp = R(N, 3), R(n, 12)
# parents: ~5 children each, children in ascending order of a random parent (the level plan's order is data dependent)
pos = torch.sort(torch.randint(0, n_par, (n,), device=dev))[0][torch.randperm(n, device=dev)]
a_rows = torch.randint(0, N, (n,), device=dev)
base_f, base_s = R(n_par, 50), R(n_par, 6)
W1, b1, W2q, b2q = R(100, in_dim) * 0.2, R(100) * 0.1, R(3, 100) * 0.1, R(3) * 0.1
xf, xs, xo = R(N, 50), R(N, 6), R(N, 30)
rows = torch.randperm(N, device=dev)[:n].contiguous()
X = torch.empty(n, in_dim, device=dev)
yf, ys, yo, Q = (torch.empty(n, w, device=dev) for w in (50, 6, 30, 3))
sums = torch.zeros(int(L.cgs_means_accum_doubles()), dtype=torch.float64, device=dev)
dyf, dys, dyo = R(n, 50), R(n, 6), R(n, 30)
m = int(0.15 * n)
sub = torch.randperm(n, device=dev)[:m]
smap = torch.full((n,), -1, dtype=torch.int32, device=dev)
smap[sub] = torch.arange(m, dtype=torch.int32, device=dev)
sf, ss, so, sQ, dxsub = R(m, 50), R(m, 6), R(m, 30), R(m, 3), R(m, in_dim)
dxf, dxs, dxo = (torch.empty(N, w, device=dev) for w in (50, 6, 30))
dX = torch.empty(n, in_dim, device=dev)
dW1, db1, dW2q, db2q = torch.zeros(100, in_dim, device=dev), torch.zeros(100, device=dev), torch.zeros(3, 100, device=dev), torch.zeros(3, device=dev)
ws = torch.empty(int(L.cgs_ctx_level_bwd_scratch_bytes()), dtype=torch.uint8, device=dev)
st = _lib.current_stream()
seed, q0 = 12345, (1.0, 0.001, 0.2)
ctxl = in_dim == 71


def fwd():
    _lib.check(L.cgs_ctx_level_fwd(in_dim, p(anchor), N, p(a_rows), None, p(base_f) if ctxl else None, p(base_s) if ctxl else None,
                                   n_par if ctxl else 0, p(pos) if ctxl else None, p(hyp), n, p(W1), p(b1), p(W2q), p(b2q), p(xf), p(xs), p(xo), p(rows), seed,
                                   *q0, p(X), p(yf), p(ys), p(yo), p(Q), p(sums), st), "fwd")


def bwd():
    _lib.check(L.cgs_ctx_level_bwd(in_dim, p(X), p(W1), p(b1), p(W2q), p(b2q), p(dyf), p(dys), p(dyo), None, n, seed, *q0, p(rows),
                                   N, p(dxf), p(dxs), p(dxo), p(smap), m, p(sf), p(ss), p(so), p(sQ), p(dxsub), p(dX), p(dW1), p(db1),
                                   p(dW2q), p(db2q), p(ws), ws.numel(), st), "bwd")


def timeit(f, name):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / iters
    print(f"{name:28s} {us:9.1f} us  ({n} rows, in_dim {in_dim})")
    return us


t_f = timeit(fwd, "ctx_level_fwd")
t_b = timeit(bwd, "ctx_level_bwd (+reduce)")
tiles = (n + 15) // 16
mf_f = 140 if ctxl else 28
mf_b = 420 if ctxl else 84
floor = lambda mf: tiles * mf * 32 / 1024 / 2.4e3
print(f"MFMA floors (2.4 GHz, 1024 SIMDs): fwd {floor(mf_f):.1f} us, bwd {floor(mf_b):.1f} us")
print(f"bytes/row fwd ~{300 + in_dim * 4 + 344 * 2 + 20}, bwd ~{in_dim * 8 + 344 * 2 + 0.15 * (344 + in_dim * 4) + 16:.0f}")
