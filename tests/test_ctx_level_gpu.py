"""The fused level kernels (csrc/ctx_level.hip: cgs_ctx_level_fwd / cgs_ctx_level_bwd, one launch per level and direction
for the every-row half of scene/gaussian_model.py:1594-1616) against the launches they replace (rowcat -> mlp2 -> noise_quant,
noise_quant_bwd -> mlp2_bwd_rc -> wgrad), on the same model, the same noise seeds and the same rate subset.

Reference parity of the fused path itself (values and every gradient against the reference's own run) is
tests/test_training_parity_gpu.py, which drives whatever context_model.LEVEL_FUSED selects — the fused kernels by default.
"""
import itertools

import numpy as np
import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _model(N, seed):
    from contextgs_amd.model import GaussianModel
    pc = GaussianModel(feat_dim=gi.D, n_offsets=gi.K, voxel_size=0.01, level_num=gi.LEVELS, target_ratio=0.2)
    sd = pc.state_dict()
    for k, v in gi.mlp_weights(seed, positive_scales=True).items():
        sd[k] = T(v)
    pc.load_state_dict(sd, strict=False)
    st = gi.anchor_state(N, seed)
    pc.set_state(st["anchor"], st["offset"], st["mask"], st["feat"], st["hyper"], st["scaling"])
    pc.update_anchor_bound()
    pc.train()
    return pc


def _run(N, seed, fused, monkeypatch, rate_side=True):
    from contextgs_amd import context_model as cm
    from contextgs_amd import ctx_ops
    monkeypatch.setattr(cm, "LEVEL_FUSED", fused)
    monkeypatch.setattr(cm, "RATE_SIDE", rate_side)
    counter = itertools.count(1)
    monkeypatch.setattr(ctx_ops, "next_seed", lambda: 0x1234567 * next(counter) + 99)
    g = torch.Generator(device="cpu").manual_seed(seed)
    choose = (torch.rand(N, generator=g) < 0.15).cuda()
    monkeypatch.setattr(cm, "choose_mask_provider", lambda anchor, mab: choose & mab if mab is not None else choose)
    pc = _model(N, seed)
    binary, mab = pc.get_mask_pair()
    res = cm.multi_scale_generating(pc, pc.get_anchor, pc._hyper_latent, pc._anchor_feat, pc._offset, pc.get_scaling,
                                    binary, mab, training=True, predict_bpp=True)
    fq, sq, oq, bpp, bf, bs, bo, each = res
    rng = np.random.default_rng(seed + 5)
    ws = [T(rng.normal(size=tuple(t.shape)).astype(np.float32)) for t in (fq, sq, oq)]
    loss = sum((t * w).sum() for t, w in zip((fq, sq, oq), ws)) + 3000.0 * bpp + 200.0 * bf + 50.0 * bs + 70.0 * bo
    loss.backward()
    grads = {n_: p.grad.detach().clone() for n_, p in pc.named_parameters() if p.grad is not None}
    return dict(fq=fq.detach(), sq=sq.detach(), oq=oq.detach(), bits=torch.stack([bpp, bf, bs, bo]).detach(), each=list(each),
                grads=grads, loss=float(loss))


def _close(name, a, b, rtol, atol_of_max):
    a, b = a.double(), b.double()
    big = max(float(b.abs().max()), 1e-30)
    err = (a - b).abs()
    tol = rtol * b.abs() + atol_of_max * big
    frac = float((err > tol).double().mean())
    print(f"{name:34s} max|ref| {big:10.4g} max err / max {float(err.max()) / big:9.3g} outside {frac:.2e}")
    return frac, float(err.max()) / big


@pytest.mark.parametrize("N,seed,rate_side", [(3000, 2, True), (10000, 4, True), (50021, 7, True), (3000, 3, False)])
def test_fused_level_equals_the_separate_launches(N, seed, rate_side, monkeypatch):
    old = _run(N, seed, False, monkeypatch, rate_side)
    new = _run(N, seed, True, monkeypatch, rate_side)
    # same noise, same rate subset; the step sizes come from a VALU dot product instead of a padded MFMA tile: round-off
    for k in ("fq", "sq", "oq"):
        frac, worst = _close(k, new[k], old[k], 1e-5, 1e-6)
        assert frac == 0.0, (k, frac, worst)
    assert torch.allclose(new["bits"], old["bits"], rtol=2e-5), (new["bits"], old["bits"])
    assert np.allclose(new["each"][:2], old["each"][:2], rtol=1e-4)
    assert abs(new["loss"] - old["loss"]) <= 2e-5 * abs(old["loss"])
    assert set(new["grads"]) == set(old["grads"])
    for name in sorted(old["grads"]):
        # gradients carry 1 / likelihood terms of fp32 CDF differences: compare at 2e-4 of the tensor maximum with a
        # small allowance, as the rest of the suite does (the strict judgement against the fp64 reference run is
        # test_training_parity_gpu.py)
        frac, worst = _close(name, new["grads"][name], old["grads"][name], 1e-3, 2e-4)
        assert frac <= 2e-3 and worst <= 2e-2, (name, frac, worst)


def test_fused_level_is_the_default_and_runs_the_level_kernels(monkeypatch):
    """The product path launches cgs_ctx_level_fwd / _bwd (not the round-4 launches) when nothing is overridden."""
    from contextgs_amd import context_model as cm
    from contextgs_amd import ctx_ops
    assert cm.LEVEL_FUSED
    calls = {"fwd": 0, "bwd": 0}
    fwd0, bwd0 = ctx_ops._LevelFused.forward, ctx_ops._LevelFused.backward

    def fwd(*a, **k):
        calls["fwd"] += 1
        return fwd0(*a, **k)

    def bwd(*a, **k):
        calls["bwd"] += 1
        return bwd0(*a, **k)

    monkeypatch.setattr(ctx_ops._LevelFused, "forward", staticmethod(fwd))
    monkeypatch.setattr(ctx_ops._LevelFused, "backward", staticmethod(bwd))
    pc = _model(3000, 2)
    binary, mab = pc.get_mask_pair()
    res = cm.multi_scale_generating(pc, pc.get_anchor, pc._hyper_latent, pc._anchor_feat, pc._offset, pc.get_scaling,
                                    binary, mab, training=True, predict_bpp=True)
    (res[0].sum() + res[3]).backward()
    assert calls == {"fwd": gi.LEVELS, "bwd": gi.LEVELS}, calls


@pytest.mark.parametrize("n,in_dim", [(1, 15), (17, 71), (257, 71), (4099, 15)])
def test_level_kernels_against_a_torch_statement(n, in_dim):
    """cgs_ctx_level_fwd / _bwd through the C ABI on random operands vs the same maths in torch (fp64 accumulate)."""
    import ctypes as C
    from contextgs_amd import _lib, ctx_ops
    from oracle.context_ref import ctx_noise
    L = _lib.lib()
    dev = "cuda"
    gen = torch.Generator(device="cpu").manual_seed(n + in_dim)
    R = lambda *s: torch.randn(*s, generator=gen).to(dev)
    N, n_par = n + 11, max(1, n // 3)
    anchor, hyp = R(N, 3), R(n, 12)
    a_rows = torch.randperm(N, generator=gen)[:n].to(dev)
    a_mask = (torch.rand(N, generator=gen) < 0.7).to(dev)
    base_f, base_s = R(n_par, 50), R(n_par, 6)
    pos = torch.randint(0, n_par, (n,), generator=gen).to(dev)
    W1, b1 = R(100, in_dim) * 0.2, R(100) * 0.1
    W2q, b2q = R(3, 100) * 0.1, R(3) * 0.1
    xf, xs, xo = R(N, 50), R(N, 6), R(N, 30)
    rows = torch.randperm(N, generator=gen)[:n].to(dev)
    seed, q0 = 0x5DEECE66D, (1.0, 0.001, 0.2)
    X = torch.empty(n, in_dim, device=dev)
    yf, ys, yo, Q = (torch.empty(n, w, device=dev) for w in (50, 6, 30, 3))
    sums = torch.zeros(int(L.cgs_means_accum_doubles()), dtype=torch.float64, device=dev)
    p = _lib.ptr
    ctxl = in_dim == 71
    mask_u8 = a_mask.view(torch.uint8)
    _lib.check(L.cgs_ctx_level_fwd(in_dim, p(anchor), N, p(a_rows), None if ctxl else p(mask_u8), p(base_f) if ctxl else None,
                                   p(base_s) if ctxl else None, n_par if ctxl else 0, p(pos) if ctxl else None, p(hyp), n, p(W1), p(b1), p(W2q), p(b2q),
                                   p(xf), p(xs), p(xo), p(rows), seed, *q0, p(X), p(yf), p(ys), p(yo), p(Q), p(sums),
                                   _lib.current_stream()), "fwd")
    if ctxl:
        Xr = torch.cat([anchor[a_rows], base_f[pos], base_s[pos], hyp], 1)
    else:
        Xr = torch.cat([anchor[a_rows] * a_mask[a_rows].float()[:, None], hyp], 1)
    assert torch.equal(X, Xr)
    H = torch.relu(Xr.double() @ W1.double().t() + b1.double())
    qa = H @ W2q.double().t() + b2q.double()
    q0t = torch.tensor(q0, dtype=torch.float64, device=dev)
    Qr = (q0t * (1 + torch.tanh(qa))).clamp_min(1e-9)
    assert torch.allclose(Q.double() / q0t, Qr / q0t, rtol=1e-4, atol=2e-6)      # (1 + tanh) cancels near tanh = -1
    u = [torch.from_numpy(ctx_noise(seed, k, n * w).reshape(n, w)).to(dev) for k, w in enumerate((50, 6, 30))]
    for y, x, uu, k in ((yf, xf, u[0], 0), (ys, xs, u[1], 1), (yo, xo, u[2], 2)):
        assert torch.equal(y, x[rows] + uu * Q[:, k:k + 1])
    tot = sums.view(-1, 16)[:, :3].sum(0)
    # (per-lane fp32 partial sums over the lane's tiles, then double: the accumulation of noise_quant_fwd_kernel)
    assert torch.allclose(tot, torch.stack([xf[rows].double().sum(), xs[rows].double().sum(), xo[rows].double().sum()]), rtol=1e-5,
                          atol=1e-3)
    # ---- backward ----
    dyf, dys, dyo, dQe = R(n, 50), R(n, 6), R(n, 30), R(n, 3)
    m = max(1, n // 5)
    sub = torch.randperm(n, generator=gen)[:m].to(dev)
    smap = torch.full((n,), -1, dtype=torch.int32, device=dev)
    smap[sub] = torch.arange(m, dtype=torch.int32, device=dev)
    sf, ss, so, sQ, dxsub = R(m, 50), R(m, 6), R(m, 30), R(m, 3), R(m, in_dim)
    dxf, dxs, dxo = (torch.full((N, w), 7.0, device=dev) for w in (50, 6, 30))
    dX = torch.empty(n, in_dim, device=dev)
    dW1, db1, dW2q, db2q = torch.ones(100, in_dim, device=dev), torch.ones(100, device=dev), torch.ones(3, 100, device=dev), \
        torch.ones(3, device=dev)
    ws = torch.empty(int(L.cgs_ctx_level_bwd_scratch_bytes()), dtype=torch.uint8, device=dev)
    _lib.check(L.cgs_ctx_level_bwd(in_dim, p(X), p(W1), p(b1), p(W2q), p(b2q), p(dyf), p(dys), p(dyo), p(dQe), n, seed, *q0, p(rows),
                                   N, p(dxf), p(dxs), p(dxo), p(smap), m, p(sf), p(ss), p(so), p(sQ), p(dxsub), p(dX), p(dW1), p(db1),
                                   p(dW2q), p(db2q), p(ws), ws.numel(), _lib.current_stream()), "bwd")
    # cgs_ctx_level_bwd2: the same launch + the hyper columns of dX (the last 12) scattered to rows rows[r] of a [N, 12] buffer; every
    # other output bit-equal to cgs_ctx_level_bwd's, rows no level row names untouched
    dxf2, dxs2, dxo2 = (torch.full((N, w), 7.0, device=dev) for w in (50, 6, 30))
    dX2, dh = torch.empty(n, in_dim, device=dev), torch.full((N, 12), 9.0, device=dev)
    dW1b, db1b, dW2qb, db2qb = torch.ones(100, in_dim, device=dev), torch.ones(100, device=dev), torch.ones(3, 100, device=dev), \
        torch.ones(3, device=dev)
    _lib.check(L.cgs_ctx_level_bwd2(in_dim, p(X), p(W1), p(b1), p(W2q), p(b2q), p(dyf), p(dys), p(dyo), p(dQe), n, seed, *q0, p(rows),
                                    N, p(dxf2), p(dxs2), p(dxo2), p(smap), m, p(sf), p(ss), p(so), p(sQ), p(dxsub), p(dX2), p(dh), p(dW1b),
                                    p(db1b), p(dW2qb), p(db2qb), p(ws), ws.numel(), _lib.current_stream()), "bwd2")
    assert torch.equal(dX2, dX) and torch.equal(dxf2, dxf) and torch.equal(dxs2, dxs) and torch.equal(dxo2, dxo)
    assert torch.equal(dW1b, dW1) and torch.equal(db1b, db1) and torch.equal(dW2qb, dW2q) and torch.equal(db2qb, db2q)
    assert torch.equal(dh[rows], dX[:, in_dim - 12:])
    not_named = torch.ones(N, dtype=torch.bool, device=dev)
    not_named[rows] = False
    assert bool((dh[not_named] == 9.0).all())
    gf, gs, go = dyf.clone(), dys.clone(), dyo.clone()
    gf[sub] += sf; gs[sub] += ss; go[sub] += so
    for d, gg, w in ((dxf, gf, 50), (dxs, gs, 6), (dxo, go, 30)):
        assert torch.equal(d[rows], gg)
        untouched = torch.ones(N, dtype=torch.bool, device=dev)
        untouched[rows] = False
        assert bool((d[untouched] == 7.0).all())
    gQ = dQe.double().clone()
    gQ[sub] += sQ.double()
    gQ += torch.stack([(gf.double() * u[0].double()).sum(1), (gs.double() * u[1].double()).sum(1),
                       (go.double() * u[2].double()).sum(1)], 1)
    t = torch.tanh(qa)
    dq = torch.where(q0t * (1 + t) >= 1e-9, gQ * q0t * (1 - t * t), torch.zeros_like(gQ))
    dH = dq @ W2q.double()
    dZ1 = dH * (H > 0)
    dXr = dZ1 @ W1.double()
    dXr[sub] += dxsub.double()
    close = lambda a, b, tol=1e-4: float((a.double() - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-30)
    # (a hidden unit whose pre-activation is within fp32 round-off of 0 may take the other ReLU branch than the fp64 statement)
    errX = (dX.double() - dXr).abs() / max(float(dXr.abs().max()), 1e-30)
    assert float((errX > 2e-5).double().mean()) <= 1e-3 and float(errX.max()) <= 5e-2, float(errX.max())
    assert close(dW1 - 1, dZ1.t() @ Xr.double())
    assert close(db1 - 1, dZ1.sum(0))
    assert close(dW2q - 1, dq.t() @ H)
    assert close(db2q - 1, dq.sum(0))
