"""Fused context-model stages (csrc/ctx.hip) against the torch composition they replace
(scene/gaussian_model.py:1594-1616, :1650-1669 of the reference)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda")


@pytest.mark.parametrize("n,rows", [(1, 5), (1000, 300), (70001, 9000)])
def test_rowcat_matches_cat_of_gathers(n, rows):
    from contextgs_amd.ctx_ops import rowcat
    g = torch.Generator(device="cuda").manual_seed(n)
    a = torch.randn(rows, 3, device=_dev(), generator=g, requires_grad=True)
    f = torch.randn(rows, 50, device=_dev(), generator=g, requires_grad=True)
    s = torch.randn(rows, 6, device=_dev(), generator=g, requires_grad=True)
    h = torch.randn(n, 12, device=_dev(), generator=g, requires_grad=True)
    idx = torch.randint(0, rows, (n,), device=_dev(), generator=g)
    out = rowcat([(a, idx, False), (f, idx, False), (s, idx, False), (h, None, True)])
    ref = torch.cat([a[idx], f[idx], s[idx], h], dim=1)
    assert torch.equal(out, ref)                                   # pure data movement: bit exact
    w = torch.randn_like(out)
    ga = torch.autograd.grad((out * w).sum(), [a, f, s, h])
    gr = torch.autograd.grad((ref * w).sum(), [a, f, s, h])
    for x, y in zip(ga, gr):
        torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-5)     # atomic vs sorted accumulation order


def test_rowcat_distinct_rows_and_no_grad_sources():
    from contextgs_amd.ctx_ops import rowcat
    a = torch.randn(100, 3, device=_dev(), requires_grad=True)
    h = torch.randn(40, 12, device=_dev())
    idx = torch.randperm(100, device=_dev())[:40]
    out = rowcat([(a, idx, True), (h, None, True)])
    assert torch.equal(out, torch.cat([a[idx], h], 1))
    (ga,) = torch.autograd.grad(out.sum(), [a])
    ref = torch.zeros_like(a)
    ref[idx] = 1
    assert torch.equal(ga, ref)
    # empty level
    e = rowcat([(a, idx[:0], True), (h[:0], None, True)])
    assert e.shape == (0, 15)


@pytest.mark.parametrize("n", [1, 37, 50000])
def test_noise_quant(n):
    from contextgs_amd.ctx_ops import noise_quant
    g = torch.Generator(device="cuda").manual_seed(7)
    xf = torch.randn(n, 50, device=_dev(), generator=g, requires_grad=True)
    xs = torch.randn(n, 6, device=_dev(), generator=g, requires_grad=True)
    xo = torch.randn(n, 30, device=_dev(), generator=g, requires_grad=True)
    qadj = (torch.randn(n, 3, device=_dev(), generator=g) * 2).requires_grad_()
    q0 = (1.0, 0.001, 0.2)
    yf, ys, yo, Q = noise_quant(xf, xs, xo, qadj, q0, seed=1234)
    Qref = torch.stack([(q0[k] * (1 + torch.tanh(qadj[:, k]))).clamp(1e-9) for k in range(3)], 1)
    torch.testing.assert_close(Q, Qref, rtol=2e-6, atol=1e-12)
    us = [((y - x) / Q[:, k:k + 1]).detach() for k, (x, y) in enumerate(((xf, yf), (xs, ys), (xo, yo)))]
    for k, (u, x) in enumerate(zip(us, (xf, xs, xo))):
        # u is rebuilt as (y - x) / Q: allow the fp32 cancellation error of that subtraction
        slack = 2e-7 * (x.detach().abs() + 1) / Q[:, k:k + 1].detach() + 1e-6
        assert bool(((u.abs() - 0.5) <= slack).all())
    if n >= 50000:
        u = us[0]
        assert abs(u.mean().item()) < 3e-3 and abs(u.var().item() - 1 / 12) < 3e-3
        # rows / columns / tensors are decorrelated
        assert abs((u[:, 0] * u[:, 1]).mean().item()) < 3e-3
        assert abs((us[0][:, :6] * us[1]).mean().item()) < 3e-3
    # same seed -> same noise, other seed -> other noise
    yf2, *_ = noise_quant(xf, xs, xo, qadj, q0, seed=1234)
    assert torch.equal(yf, yf2)
    yf3, *_ = noise_quant(xf, xs, xo, qadj, q0, seed=1235)
    assert not torch.equal(yf, yf3)
    # gradients against the torch composition with the same (recovered) noise; u is rebuilt from y, so compare
    # loosely where Q is tiny
    wf, ws, wo, wq = torch.randn_like(yf), torch.randn_like(ys), torch.randn_like(yo), torch.randn_like(Q)
    loss = (yf * wf).sum() + (ys * ws).sum() + (yo * wo).sum() + (Q * wq).sum()
    got = torch.autograd.grad(loss, [xf, xs, xo, qadj])
    ref_loss = ((xf + us[0] * Qref[:, 0:1]) * wf).sum() + ((xs + us[1] * Qref[:, 1:2]) * ws).sum() + \
               ((xo + us[2] * Qref[:, 2:3]) * wo).sum() + (Qref * wq).sum()
    ref = torch.autograd.grad(ref_loss, [xf, xs, xo, qadj])
    for k in range(3):
        assert torch.equal(got[k], ref[k])
    torch.testing.assert_close(got[3], ref[3], rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("n_l,frac,clamp", [(1, 1.0, True), (2000, 0.15, True), (2000, 0.3, False), (5, 0.0, True)])
def test_level_rate(n_l, frac, clamp):
    from contextgs_amd import encodings
    from contextgs_amd.ctx_ops import level_rate
    from contextgs_amd.entropy_models import Entropy_gaussian
    D, K, N = 50, 10, 3000
    g = torch.Generator(device="cuda").manual_seed(n_l)
    R = lambda *s: torch.randn(*s, device=_dev(), generator=g)
    yf, ys, yo = R(n_l, D).requires_grad_(), (R(n_l, 6) * 0.01).requires_grad_(), R(n_l, 3 * K).requires_grad_()
    Q = torch.stack([torch.rand(n_l, device=_dev(), generator=g) + 0.5,
                     torch.rand(n_l, device=_dev(), generator=g) * 0.002 + 1e-4,
                     torch.rand(n_l, device=_dev(), generator=g) * 0.3 + 0.05], 1).requires_grad_()
    loc = torch.nonzero(torch.rand(n_l, device=_dev(), generator=g) < frac)[:, 0]
    n_sub = loc.shape[0]
    pred = R(n_sub, 175)
    E = D + 6 + 3 * K
    with torch.no_grad():   # positive-ish scales, small ones for the scaling block
        pred[:, D:2 * D] = pred[:, D:2 * D].abs() + 0.1
        pred[:, 2 * D:2 * D + 6] *= 0.01
        pred[:, 2 * D + 6:2 * D + 12] = pred[:, 2 * D + 6:2 * D + 12].abs() * 0.01 + 1e-3
        pred[:, 2 * D + 12 + 3 * K:2 * E] = pred[:, 2 * D + 12 + 3 * K:2 * E].abs() + 0.1
    pred.requires_grad_()
    masks = (torch.rand(N, K, device=_dev(), generator=g) < 0.6).float().requires_grad_()
    grows = torch.randint(0, N, (n_sub,), device=_dev(), generator=g)
    x_means = torch.tensor([0.1, 0.0, -0.05], device=_dev())
    old = encodings.use_clamp
    encodings.use_clamp = clamp
    try:
        sums = level_rate(yf, ys, yo, Q, pred, loc, masks, grows, x_means, clamp, K)
        eg = Entropy_gaussian(Q=1)
        mf, sf, ms, ss, mo, so, _ = torch.split(pred, [D, D, 6, 6, 3 * K, 3 * K, 3], dim=1)
        bf = eg(yf[loc], mf, sf, Q[loc, 0:1], x_means[0])
        bs = eg(ys[loc], ms, ss, Q[loc, 1:2], x_means[1])
        bo = eg(yo[loc], mo, so, Q[loc, 2:3], x_means[2]) * masks[grows].repeat_interleave(3, dim=1)
        ref = torch.stack([bf.sum(), bs.sum(), bo.sum()])
        torch.testing.assert_close(sums, ref, rtol=2e-5, atol=1e-3)
        w = torch.tensor([1.0, 0.5, 2.0], device=_dev())
        got = torch.autograd.grad((sums * w).sum(), [yf, ys, yo, Q, pred, masks], allow_unused=True)
        exp = torch.autograd.grad((ref * w).sum(), [yf, ys, yo, Q, pred, masks], allow_unused=True)
        for a, b in zip(got, exp):
            if b is None:
                assert a is None or not a.abs().any()
                continue
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    finally:
        encodings.use_clamp = old


@pytest.mark.parametrize("n_par,n_child", [(1, 1), (500, 2000), (20000, 81234)])
def test_ctx_assemble_csr_backward_equals_scatter_add(n_par, n_child):
    """cgs_ctx_gather_bwd (per-parent sums over the plan's child lists) == the atomic scatter-add backward."""
    from contextgs_amd.ctx_ops import ctx_assemble, rowcat
    g = torch.Generator(device="cuda").manual_seed(n_child)
    N = n_par * 3 + 7
    anchor = torch.randn(N, 3, device=_dev(), generator=g, requires_grad=True)
    base_f = torch.randn(n_par, 50, device=_dev(), generator=g, requires_grad=True)
    base_s = torch.randn(n_par, 6, device=_dev(), generator=g, requires_grad=True)
    own = torch.randn(n_child, 12, device=_dev(), generator=g, requires_grad=True)
    prow = torch.randperm(N, device=_dev(), generator=g)[:n_par]                 # original row of every coded parent
    pos = torch.randint(0, n_par, (n_child,), device=_dev(), generator=g)
    if n_par > 10:
        pos[pos == 3] = 4                                                        # a parent without children
    idx = prow[pos]
    order = torch.argsort(pos, stable=True)
    offs = torch.cat([torch.zeros(1, dtype=torch.long, device=_dev()), torch.bincount(pos, minlength=n_par).cumsum(0)])
    out = ctx_assemble(anchor, base_f, base_s, own, idx, pos, (offs, order, prow))
    ref = rowcat([(anchor, idx, False), (base_f, pos, False), (base_s, pos, False), (own, None, True)])
    assert torch.equal(out, ref)
    w = torch.randn_like(out)
    got = torch.autograd.grad((out * w).sum(), [anchor, base_f, base_s, own])
    exp = torch.autograd.grad((ref * w).sum(), [anchor, base_f, base_s, own])
    for a, b in zip(got, exp):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("complete", [True, False])
def test_noise_quant_row_source_equals_gather_then_noise_quant(complete):
    """RowSource path (the level kernels read rows of the parameter tensors through perm[lo:hi] and scatter the
    gradients back) == gather into coding order, split, noise_quant per level, cat, index_copy: same outputs with
    the same seeds, same parameter gradients bit for bit, incl. a row set that does not cover every row."""
    from contextgs_amd import ctx_ops as ops
    torch.manual_seed(3)
    N, D, S, K = 700, 50, 6, 10
    dev = "cuda"
    perm_all = torch.randperm(N, device=dev)
    perm = perm_all if complete else perm_all[: N - 90]
    sizes = [120, 0, perm.shape[0] - 120]
    mk = lambda *s: torch.randn(*s, device=dev)
    base = [mk(N, D), mk(N, S), mk(N, K, 3)]
    qadjs = [mk(n, 3) for n in sizes]
    wf, ws, wo = mk(perm.shape[0], D), mk(perm.shape[0], S), mk(perm.shape[0], 3 * K)
    q0 = (1.0, 0.001, 0.2)

    def run(row_source):
        feat, scal, off = (t.clone().requires_grad_(True) for t in base)
        qs = [q.clone().requires_grad_(True) for q in qadjs]
        outs, lo = [], 0
        if row_source:
            src = ops.RowSource(feat, scal, off, complete)
            for j, n in enumerate(sizes):
                outs.append(ops.noise_quant(None, None, None, qs[j], q0, seed=100 + j, src=src, rows=perm[lo:lo + n]))
                lo += n
        else:
            fp, sp, op = feat[perm], scal[perm], off[perm].reshape(-1, 3 * K)
            for j, n in enumerate(sizes):
                outs.append(ops.noise_quant(fp[lo:lo + n], sp[lo:lo + n], op[lo:lo + n], qs[j], q0, seed=100 + j))
                lo += n
        yf, ys, yo = (torch.cat([o[t] for o in outs]) for t in range(3))
        Q = torch.cat([o[3] for o in outs])
        loss = (yf * wf).sum() + (ys * ws).sum() + (yo * wo).sum() + (Q * Q).sum()
        loss.backward()
        return (yf, ys, yo, Q), (feat.grad, scal.grad, off.grad), [q.grad for q in qs]

    a, b = run(True), run(False)
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x, y)
    for x, y in zip(a[1], b[1]):
        assert x.shape == y.shape and torch.equal(x, y)
    for x, y in zip(a[2], b[2]):
        assert torch.equal(x, y)


def test_mask_ste_equals_the_torch_accessors():
    """cgs_mask_ste_{fwd,bwd} == get_mask / get_mask_anchor as the reference writes them (:295-310): values bit for
    bit (same fp32 expression), gradient = sigmoid backward."""
    from contextgs_amd.ctx_ops import mask_ste
    torch.manual_seed(0)
    m = (torch.randn(5000, 10, 1, device="cuda") * 4 - 2).requires_grad_(True)
    with torch.no_grad():
        m[:40] = -20.0                                   # dead anchors
        m[40:80, 3] = torch.log(torch.tensor(0.01 / 0.99)) + torch.linspace(-1e-3, 1e-3, 40, device="cuda")[:, None]  # at the threshold
    s = torch.sigmoid(m)
    ref_mask = ((s > 0.01).float() - s).detach() + s
    ref_any = torch.sum(ref_mask, dim=1)[:, 0] > 0
    w = torch.randn_like(m)
    (g_ref,) = torch.autograd.grad((ref_mask * w).sum(), m)
    mask, alive = mask_ste(m)
    assert torch.equal(alive, ref_any) and not bool(alive[:40].any())
    assert torch.equal(mask, ref_mask)
    (g,) = torch.autograd.grad((mask * w).sum(), m)
    assert torch.allclose(g, g_ref, rtol=1e-6, atol=1e-9)


def test_means3_equals_torch_means():
    from contextgs_amd.ctx_ops import means3
    torch.manual_seed(1)
    a, b, c = torch.randn(40000, 50, device="cuda") + 0.3, torch.randn(40000, 6, device="cuda") * 0.5 - 4, torch.randn(40000, 10, 3, device="cuda") * 0.2
    got = means3(a, b, c, exp_b=True)
    want = torch.stack([a.double().mean(), torch.exp(b).double().mean(), c.double().mean()]).float()
    assert torch.allclose(got, want, rtol=2e-6, atol=1e-8)
    assert torch.equal(means3(a, b, c, exp_b=True), got)                      # deterministic
    got2 = means3(a[:1], b[:0], c, exp_b=False)
    assert torch.allclose(got2[0], a[:1].mean()) and float(got2[1]) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("n,with_perm,with_mask", [(1, False, False), (4097, True, True), (100_003, True, False), (50_000, False, True)])
def test_choose_rows_matches_the_torch_composition(n, with_perm, with_mask):
    """ctx_plan.hip (cgs_ctx_choose_flags / _compact): chosen rows in coding order, their original indices, level-local
    positions, per-level counts, live count, the inverse map per level and the plan-validity flag, against the torch
    statement of scene/gaussian_model.py:1658-1661 restricted to the levels."""
    from contextgs_amd import ctx_ops
    dev = "cuda"
    g = torch.Generator().manual_seed(n)
    perm = torch.randperm(n, generator=g).to(dev) if with_perm else None
    mask = (torch.rand(n, generator=g) < 0.8).to(dev) if with_mask else None
    given = (torch.rand(n, generator=g) < 0.15).to(dev)
    cuts = sorted(set([0, n // 7, n // 3, n]))
    anchor = torch.randn(n, 3, generator=g).to(dev)
    stale, live, per_level, nz, rows, loc, sub_map = ctx_ops.choose_rows(perm, n, mask, given, 0, 0.15, anchor, anchor.clone(),
                                                                          None if mask is None else mask.clone(), cuts)
    order = perm if perm is not None else torch.arange(n, device=dev)
    flag = given[order] & (mask[order] if mask is not None else torch.ones(n, dtype=torch.bool, device=dev))
    want_nz = torch.nonzero(flag)[:, 0]
    assert not stale and live == (int(mask.sum()) if mask is not None else n)
    assert torch.equal(nz, want_nz) and torch.equal(rows, order[want_nz])
    lvl = torch.bucketize(want_nz, torch.tensor(cuts[1:-1], device=dev), right=True) if len(cuts) > 2 else torch.zeros_like(want_nz)
    starts = torch.tensor(cuts[:-1], device=dev)
    assert torch.equal(loc, want_nz - starts[lvl])
    assert per_level == [int((lvl == l).sum()) for l in range(len(cuts) - 1)]
    # inverse map: row r of level l -> index in l's part of the chosen list
    want_map = torch.full((n,), -1, dtype=torch.int32, device=dev)
    cum = [0]
    for c in per_level:
        cum.append(cum[-1] + c)
    for l in range(len(cuts) - 1):
        sel = want_nz[cum[l]:cum[l + 1]]
        want_map[sel] = torch.arange(sel.numel(), dtype=torch.int32, device=dev)
    if sum(per_level):
        assert torch.equal(sub_map[:n], want_map)
    # the counter-based draw is used when no mask is given: ~15 % of the live rows, reproducible by seed
    _s, _l, pl1, nz1, _r, _lo, _m = ctx_ops.choose_rows(perm, n, mask, None, 1234, 0.15, anchor, None, None, cuts)
    _s, _l, pl2, nz2, _r, _lo, _m = ctx_ops.choose_rows(perm, n, mask, None, 1234, 0.15, anchor, None, None, cuts)
    assert pl1 == pl2 and torch.equal(nz1, nz2)
    if n > 10_000:
        assert abs(sum(pl1) / max(1, live) - 0.15) < 0.01
    # a moved anchor / flipped mask bit is reported
    moved = anchor.clone(); moved[n // 2, 1] += 1.0
    assert ctx_ops.choose_rows(perm, n, mask, given, 0, 0.15, moved, anchor, None if mask is None else mask.clone(), cuts)[0]


@pytest.mark.parametrize("n,rows", [(1, 5), (777, 1000), (20000, 20000)])
def test_rowcat_row_mask_equals_the_product(n, rows):
    """rowcat with a per-source-row mask == cat([(src * mask[:, None])[idx], other]) — `anchor * mask_anchor.unsqueeze(1)` of
    scene/gaussian_model.py:1758-1759 folded into the level gather (cgs_rowcat_*_masked): values (incl. -0.0 for masked
    negative entries, as the product gives) and both gradients."""
    from contextgs_amd import ctx_ops
    torch.manual_seed(n)
    dev = "cuda"
    a = torch.randn(rows, 3, device=dev, requires_grad=True)
    b = torch.randn(n, 12, device=dev, requires_grad=True)
    mask = torch.rand(rows, device=dev) < 0.7
    idx = torch.randperm(rows, device=dev)[:n]
    out = ctx_ops.rowcat([(a, idx, True, mask), (b, None, True)])
    a2, b2 = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    ref = torch.cat([(a2 * mask.unsqueeze(1))[idx], b2], dim=1)
    assert torch.equal(out, ref) and torch.equal(torch.signbit(out), torch.signbit(ref))
    g = torch.randn_like(out)
    out.backward(g); ref.backward(g)
    assert torch.equal(a.grad, a2.grad) and torch.equal(b.grad, b2.grad)


@pytest.mark.parametrize("N,frac,w", [(1000, 0.995, 3), (50000, 0.9, 10), (4096, 1.0, 30), (1000, 0.0, 7), (3000, 0.3, 256),
                                       (10, 0.5, 1), (100003, 0.995, 3), (100003, 0.97, 10), (5000, 0.8, 12), (5000, 0.6, 4),
                                       (5000, 0.9, 5), (777, 0.5, 2), (1_000_000, 0.995, 10), (1_000_000, 0.995, 3)])
def test_scatter_rows_sorted_equals_zeros_index_copy(N, frac, w):
    """cgs_scatter_rows_sorted (the one-pass backward of x[visible rows]) == zeros(N, w).index_copy_(0, idx, g) for ascending idx:
    dense and sparse lists, an empty list, every row listed, the widest row the kernel takes."""
    from contextgs_amd import _lib
    torch.manual_seed(N + w)
    dev = "cuda"
    idx = torch.nonzero(torch.rand(N, device=dev) < frac)[:, 0] if 0.0 < frac < 1.0 else (
        torch.arange(N, device=dev) if frac >= 1.0 else torch.zeros(0, dtype=torch.int64, device=dev))
    n = int(idx.numel())
    g = torch.randn(n, w, device=dev)
    out = torch.full((N, w), float("nan"), device=dev)
    _lib.check(_lib.lib().cgs_scatter_rows_sorted(_lib.ptr(g), _lib.ptr(idx), n, N, w, _lib.ptr(out), _lib.current_stream()),
               "cgs_scatter_rows_sorted")
    ref = torch.zeros(N, w, device=dev)
    if n:
        ref.index_copy_(0, idx, g)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("n,rows,w", [(1, 3, 4), (1000, 5000, 12), (70001, 70001, 8), (33, 40, 64)])
def test_single_source_row_gather_with_whole_float4_rows(n, rows, w):
    """gather_rows_nograd on rows of whole float4s takes the 16-byte kernel (rowgather4_kernel): == x[idx]."""
    from contextgs_amd import ctx_ops
    torch.manual_seed(n)
    x = torch.randn(rows, w, device="cuda")
    idx = torch.randint(0, rows, (n,), device="cuda")
    assert torch.equal(ctx_ops.gather_rows_nograd(x, idx), x[idx])


@pytest.mark.parametrize("with_idx", [True, False])
def test_gather_rows_segmented_equals_cat_then_gather(with_idx):
    """cgs_gather_rows_segmented: rows of the virtual concatenation of row blocks (one contiguous, one a strided column slice,
    one missing = zeros, one empty) through an index — == cat(blocks)[idx]."""
    import ctypes as C
    from contextgs_amd import _lib
    torch.manual_seed(0)
    w = 12
    a = torch.randn(37, w, device="cuda")
    big = torch.randn(1000, 71, device="cuda")
    b = big[:, 59:71]                                         # strided block: row stride 71 floats
    sizes = [37, 1000, 50, 0]
    blocks = [a, b, None, torch.zeros(0, w, device="cuda")]
    ref_cat = torch.cat([a, b, torch.zeros(50, w, device="cuda")])
    N = sum(sizes)
    idx = torch.randperm(N, device="cuda") if with_idx else None
    begin = [0]
    for s_ in sizes:
        begin.append(begin[-1] + s_)
    out = torch.full((N, w), float("nan"), device="cuda")
    k = len(blocks)
    _lib.check(_lib.lib().cgs_gather_rows_segmented(
        k, (C.c_void_p * k)(*[None if (t is None or t.numel() == 0) else t.data_ptr() for t in blocks]),
        (C.c_int64 * k)(*[w if (t is None or t.numel() == 0) else int(t.stride(0)) for t in blocks]),
        (C.c_int64 * (k + 1))(*begin), _lib.ptr(idx), N, w, _lib.ptr(out), _lib.current_stream()), "cgs_gather_rows_segmented")
    assert torch.equal(out, ref_cat[idx] if with_idx else ref_cat)


@pytest.mark.parametrize("shape", [(5000, 1), (5000, 3), (5000, 10, 1), (5000, 10), (5000, 12), (5000, 7), (5000, 50), (3, 3)])
def test_gather_unique_forward_is_index_select(shape):
    """cgs_gather_rows (lane-per-row / float2 / float4 forms, round 4) behind context_model.gather_unique: bit-equal to
    x.index_select(0, idx) for ascending and for permuted unique indices, and the backward still scatters."""
    from contextgs_amd.context_model import gather_unique
    torch.manual_seed(sum(shape))
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    n = shape[0]
    for idx in (torch.nonzero(torch.rand(n, device="cuda") < 0.9)[:, 0], torch.randperm(n, device="cuda")[: max(1, n // 3)]):
        y = gather_unique(x, idx)
        assert torch.equal(y, x.detach().index_select(0, idx))
        g = torch.randn_like(y)
        (gx,) = torch.autograd.grad(y, x, g)
        ref = torch.zeros_like(x).index_copy_(0, idx, g)
        assert torch.equal(gx, ref)


def test_zero_unlisted_rows_equals_a_zero_fill():
    """ctx_ops.zero_unlisted_rows (cgs_mark_rows + cgs_zero_unmarked_rows): after it and a write of the listed rows, the
    buffers equal zeros().index_copy_(listed rows) — for a dense list, a sparse one, an empty one, repeated calls on the
    same stamp array, five arrays (two launches) and a reused index tensor."""
    from contextgs_amd import ctx_ops
    g = torch.Generator(device="cuda").manual_seed(4)
    n_full = 100_003
    for n in (n_full - 517, 1000, 0, n_full - 1):
        idx = torch.randperm(n_full, device="cuda", generator=g)[:n].contiguous()
        widths = [(6,), (10, 3), (50,), (1,), (3,)]
        vals = [torch.randn((n,) + w, device="cuda", generator=g) for w in widths]
        for rep in range(2):                                   # the second pass reuses the marks of the same index tensor
            bufs = [torch.full((n_full,) + w, float("nan"), device="cuda") for w in widths]
            ctx_ops.zero_unlisted_rows(idx, n_full, bufs)
            for b, v, w in zip(bufs, vals, widths):
                b[idx] = v
                ref = torch.zeros((n_full,) + w, device="cuda")
                ref[idx] = v
                assert torch.equal(b, ref)
    # an index tensor modified in place is marked again
    idx = torch.arange(10, device="cuda")
    buf = torch.full((20, 2), float("nan"), device="cuda")
    ctx_ops.zero_unlisted_rows(idx, 20, [buf])
    assert torch.isnan(buf[:10]).all() and (buf[10:] == 0).all()
    idx += 10
    buf = torch.full((20, 2), float("nan"), device="cuda")
    ctx_ops.zero_unlisted_rows(idx, 20, [buf])
    assert torch.isnan(buf[10:]).all() and (buf[:10] == 0).all()
