"""Timing of the fused rate-subset kernels (cgs_rate_sub_fwd / _bwd) against the launches they replace (row gather + cgs_mlp2_forward
+ cgs_level_rate_fwd; cgs_level_rate_bwd + cgs_mlp2_backward incl. its weight gradients) at the headline scene's level sizes.
Usage: python tools/rate_sub_micro.py [n_level m_subset in_dim]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd import _lib, mlp as _mlp

L = _lib.lib()
p = _lib.ptr
dev = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main(n, m, in_dim):
    g = torch.Generator(device="cpu").manual_seed(1)
    R = lambda *s: torch.randn(*s, generator=g).to(dev)
    X = R(n, in_dim)
    loc = torch.sort(torch.randperm(n, generator=g)[:m])[0].to(dev)
    W1, b1, W2, b2 = R(100, in_dim) * 0.2, R(100) * 0.1, R(175, 100) * 0.1, R(175) * 0.1
    b2[50:100] += 1.5; b2[106:112] += 1.5; b2[142:172] += 1.5
    yf, ys, yo = R(n, 50), R(n, 6), R(n, 30)
    Q = torch.rand(n, 3, generator=g).to(dev) * 0.5 + 0.05
    masks = (torch.rand(m, 10, generator=g) < 0.6).float().to(dev)
    xm = torch.tensor([0.1, -0.2, 0.05], device=dev)
    gs = torch.tensor([0.7, -1.3, 2.1], device=dev)
    st = _lib.current_stream()
    sums = torch.zeros(3, device=dev)
    fwd = lambda: _lib.check(L.cgs_rate_sub_fwd(in_dim, p(X), n, p(loc), m, p(W1), p(b1), p(W2), p(b2), p(yf), p(ys), p(yo), p(Q), p(masks),
                                                p(xm), 1, p(sums), st), "f")
    sf, ss, so, sq = (torch.empty(m, w, device=dev) for w in (50, 6, 30, 3))
    dx, dm = torch.empty(m, in_dim, device=dev), torch.empty(m, 10, device=dev)
    dW1, db1, dW2, db2 = (torch.empty_like(t) for t in (W1, b1, W2, b2))
    ws = torch.empty(int(L.cgs_rate_sub_bwd_scratch_bytes(in_dim, m)), dtype=torch.uint8, device=dev)
    bwd = lambda: _lib.check(L.cgs_rate_sub_bwd(in_dim, p(X), n, p(loc), m, p(W1), p(b1), p(W2), p(b2), p(yf), p(ys), p(yo), p(Q), p(masks),
                                                p(xm), 1, p(gs), p(sf), p(ss), p(so), p(sq), p(dx), p(dm), p(dW1), p(db1), p(dW2), p(db2),
                                                p(ws), ws.numel(), st), "b")
    # the launches of rounds 3-5
    x_sub = torch.empty(m, in_dim, device=dev); pred = torch.empty(m, 175, device=dev); h = torch.empty(m, 100, device=dev)
    sums2 = torch.zeros(3, device=dev)

    def old_fwd():
        torch.index_select(X, 0, loc, out=x_sub)
        _lib.check(L.cgs_mlp2_forward(in_dim, 100, 175, 0, p(x_sub), in_dim, p(W1), p(b1), p(W2), p(b2), p(pred), 175, p(h), m, st), "m")
        _lib.check(L.cgs_level_rate_fwd(p(yf), p(ys), p(yo), p(Q), p(loc), p(pred), p(masks), None, p(xm), 1, m, 50, 10, 175, p(sums2), st), "r")
    d_pred = torch.empty_like(pred); dz1 = torch.empty(m, 100, device=dev); dxs = torch.empty(m, in_dim, device=dev)
    wsw = _mlp._wgrad_workspace(torch.device(dev))
    dm2 = torch.zeros(m, 10, device=dev)

    def old_bwd():
        _lib.check(L.cgs_level_rate_bwd(p(yf), p(ys), p(yo), p(Q), p(loc), p(pred), p(masks), None, p(xm), 1, m, 50, 10, 175, p(gs), p(d_pred),
                                        p(sf), p(ss), p(so), p(sq), p(dm2), 1, st), "rb")
        _lib.check(L.cgs_mlp2_backward(in_dim, 100, 175, 0, p(x_sub), in_dim, p(W1), None, p(W2), None, p(d_pred), 175, p(h), p(dxs), in_dim, 0,
                                       p(dz1), None, p(dW1), p(db1), p(dW2), p(db2), m, p(wsw), wsw.numel(), st), "mb")
    tf, tb, of, ob = timeit(fwd), timeit(bwd), timeit(old_fwd), timeit(old_bwd)
    print(f"n {n} m {m} in {in_dim}: fused fwd {tf:7.1f} us  bwd {tb:7.1f} us | separate fwd {of:7.1f} us  bwd {ob:7.1f} us")
    print("sums fused / separate:", (sums / 23).tolist(), (sums2 / 23).tolist())


if __name__ == "__main__":
    if len(sys.argv) == 4:
        main(*map(int, sys.argv[1:]))
    else:
        for n, m, i in ((807417, 121000, 71), (154958, 23200, 71), (37625, 5640, 15)):
            main(n, m, i)
