"""Fused fp32-MFMA 2-layer MLP kernels (csrc/mlp.hip) vs a plain PyTorch fp32 reference of
the same nn.Sequential (the one floating-point op here whose oracle is torch itself).
v_mfma_f32_16x16x4_f32 is exact fp32 (an fmaf chain), so only the summation order differs:
tolerance 2e-5 relative to the tensor's max for activations, 2e-4 for the weight gradients
(sums over all rows)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

CONFIGS = [(54, 50, 10, nn.Tanh), (54, 50, 30, nn.Sigmoid), (54, 50, 70, None), (71, 100, 175, None), (15, 100, 175, None),
           (71, 100, 3, None), (15, 100, 3, None)]


def _seq(i, h, o, act):
    layers = [nn.Linear(i, h), nn.ReLU(True), nn.Linear(h, o)]
    if act is not None:
        layers.append(act())
    return nn.Sequential(*layers).cuda()


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("n", [1, 31, 32, 33, 1000, 40001])
def test_forward_backward_match_torch(cfg, n):
    from contextgs_amd import mlp
    i, h, o, act = cfg
    torch.manual_seed(n + i + o)
    seq = _seq(i, h, o, act)
    assert mlp.supported(seq)
    x = torch.randn(n, i, device="cuda", requires_grad=True)
    w = torch.randn(n, o, device="cuda")
    y = mlp.mlp2(x, seq)
    (y * w).sum().backward()
    got = [x.grad.clone()] + [p.grad.clone() for p in seq.parameters()]
    x.grad = None
    seq.zero_grad()
    y_ref = seq(x)
    (y_ref * w).sum().backward()
    ref = [x.grad] + [p.grad for p in seq.parameters()]
    assert (y - y_ref).abs().max() <= 2e-5 * max(1.0, float(y_ref.abs().max()))
    for a, b, name in zip(got, ref, ["dx", "dW1", "db1", "dW2", "db2"]):
        tol = (2e-5 if name == "dx" else 2e-4) * max(1e-6, float(b.abs().max()))
        assert (a - b).abs().max() <= tol, (name, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("cfg", [(71, 100, 3, None), (15, 100, 3, None)])
@pytest.mark.parametrize("n", [1, 17, 5000, 70001])
def test_tiny_output_backward_with_and_without_saved_hidden_layer(cfg, n, monkeypatch):
    """{71,15} -> 100 -> 3: the default backward recomputes the hidden layer (csrc/mlp_small.hip); with the knob
    off it reads the stored one (csrc/mlp.hip).  Same forward bits, gradients equal up to summation order."""
    from contextgs_amd import mlp
    i, h, o, act = cfg
    torch.manual_seed(n)
    seq = _seq(i, h, o, act)
    x = torch.randn(n, i, device="cuda", requires_grad=True)
    w = torch.randn(n, o, device="cuda")
    outs = []
    for rc in (True, False):
        monkeypatch.setattr(mlp, "RECOMPUTE_HIDDEN", rc)
        x.grad = None
        seq.zero_grad()
        y = mlp.mlp2(x, seq)
        (y * w).sum().backward()
        outs.append([y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in seq.parameters()])
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b, name in zip(outs[0][1:], outs[1][1:], ["dx", "dW1", "db1", "dW2", "db2"]):
        tol = (1e-6 if name == "dx" else 2e-4) * max(1e-6, float(b.abs().max()))
        assert (a - b).abs().max() <= tol, (name, float((a - b).abs().max()), float(b.abs().max()))


def test_empty_no_grad_and_unsupported_shape():
    from contextgs_amd import mlp
    seq = _seq(71, 100, 175, None)
    y = mlp.mlp2(torch.zeros(0, 71, device="cuda"), seq)
    assert y.shape == (0, 175)
    with torch.no_grad():
        x = torch.randn(100, 71, device="cuda")
        assert torch.allclose(mlp.mlp2(x, seq), seq(x), atol=2e-5)
    odd = _seq(33, 20, 7, None)
    assert not mlp.supported(odd)
    with pytest.raises(NotImplementedError):
        mlp.mlp2(torch.randn(4, 33, device="cuda"), odd)


@pytest.mark.parametrize("n", [1, 33, 5000, 100003])
def test_anchor_mlp3_matches_three_torch_mlps(n):
    from contextgs_amd import mlp
    torch.manual_seed(n)
    mo, mc, mv = _seq(54, 50, 10, nn.Tanh), _seq(54, 50, 30, nn.Sigmoid), _seq(54, 50, 70, None)
    assert mlp.anchor_mlp3_supported(mo, mc, mv)
    x = torch.randn(n, 54, device="cuda", requires_grad=True)
    ws = [torch.randn(n, o, device="cuda") for o in (10, 30, 70)]
    ys = mlp.anchor_mlp3(x, mo, mc, mv)
    sum((y * w).sum() for y, w in zip(ys, ws)).backward()
    params = [p for s in (mo, mc, mv) for p in s.parameters()]
    got = [x.grad.clone()] + [p.grad.clone() for p in params]
    x.grad = None
    for s in (mo, mc, mv):
        s.zero_grad()
    refs = [s(x) for s in (mo, mc, mv)]
    sum((y * w).sum() for y, w in zip(refs, ws)).backward()
    ref = [x.grad] + [p.grad for p in params]
    for y, r in zip(ys, refs):
        assert (y - r).abs().max() <= 2e-5 * max(1.0, float(r.detach().abs().max()))
    for i, (a, b) in enumerate(zip(got, ref)):
        tol = (2e-5 if i == 0 else 2e-4) * max(1e-6, float(b.abs().max()))
        assert (a - b).abs().max() <= tol, (i, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("keep_x", [True, False])
def test_anchor_mlp3_rows_equals_the_materialised_input(keep_x, monkeypatch):
    """(keep_x False: the forward does not store its assembled rows and the fused backward assembles them again — round 6's
    X-less mode, measured slower in the step and left off: profiles/r06_mlp3_xout_ab.txt.)
    cgs_anchor_mlp3_{forward,backward}_rows (input row = [feat_src[src_row] | view direction | distance] assembled in
    the kernel, gradient scattered into the source rows / pulled back to the anchors) against the same three MLPs on the
    torch-assembled [n,54] input (gaussian_renderer/__init__.py:106-110)."""
    import torch.nn as nn
    from contextgs_amd import mlp
    monkeypatch.setattr(mlp, "KEEP_X_OFF", not keep_x)
    torch.manual_seed(3)
    dev = "cuda"
    mk = lambda out, act: nn.Sequential(nn.Linear(54, 50), nn.ReLU(True), nn.Linear(50, out), *([act()] if act else [])).to(dev)
    mo, mc, mv = mk(10, nn.Tanh), mk(30, nn.Sigmoid), mk(70, None)
    n_src, n = 5000, 3777
    feat_src = torch.randn(n_src, 50, device=dev, requires_grad=True)
    src_row = torch.randperm(n_src, device=dev)[:n].contiguous()
    anchor = (torch.randn(n, 3, device=dev) * 2).requires_grad_(True)
    cam = torch.tensor([0.3, -3.0, 0.5], device=dev)
    ws = [torch.randn(n, k, device=dev) for k in (10, 30, 70)]

    def loss(outs):
        return sum((o * w).sum() for o, w in zip(outs, ws))

    params = [p for m in (mo, mc, mv) for p in m.parameters()]
    loss(mlp.anchor_mlp3_rows(feat_src, src_row, anchor, cam, mo, mc, mv)).backward()
    got = [feat_src.grad.clone(), anchor.grad.clone()] + [p.grad.clone() for p in params]
    outs_rows = [o.detach() for o in mlp.anchor_mlp3_rows(feat_src, src_row, anchor, cam, mo, mc, mv)]
    for t in [feat_src, anchor] + params:
        t.grad = None
    u = anchor - cam
    d = u.norm(dim=1, keepdim=True)
    x = torch.cat([feat_src[src_row], u / d, d], dim=1)
    outs_ref = mlp.anchor_mlp3(x, mo, mc, mv)
    loss(outs_ref).backward()
    ref = [feat_src.grad, anchor.grad] + [p.grad for p in params]
    for a, b in zip(outs_rows, outs_ref):
        assert torch.allclose(a, b.detach(), rtol=1e-5, atol=1e-6)
    for a, b in zip(got, ref):
        assert (a - b).abs().max() <= 2e-5 * max(1e-6, float(b.abs().max())), float((a - b).abs().max())
    # rows of the source that no anchor reads get exactly zero
    unread = torch.ones(n_src, dtype=torch.bool, device=dev)
    unread[src_row] = False
    assert float(got[0][unread].abs().sum()) == 0.0


@pytest.mark.parametrize("n", [1, 15, 16, 17, 3777, 100003])
@pytest.mark.parametrize("keep_x", [True, False])
def test_anchor_mlp3_rows_tiled_handover_changes_no_bit(n, keep_x, monkeypatch):
    """(round 6) Hcat in fragment-major form between the forward and the fused backward
    (cgs_anchor_mlp3_{forward,backward}_rows_t, tiled = 1) against the row-major buffers (tiled = 0): only the layout of a
    private buffer differs, so the three outputs and every gradient are the same bits — including row counts that end inside a
    16-row tile and the X-less backward."""
    import torch.nn as nn
    from contextgs_amd import mlp
    monkeypatch.setattr(mlp, "KEEP_X_OFF", not keep_x)
    torch.manual_seed(5)
    dev = "cuda"
    mk = lambda out, act: nn.Sequential(nn.Linear(54, 50), nn.ReLU(True), nn.Linear(50, out), *([act()] if act else [])).to(dev)
    mo, mc, mv = mk(10, nn.Tanh), mk(30, nn.Sigmoid), mk(70, None)
    n_src = n + 1234
    feat_src = torch.randn(n_src, 50, device=dev, requires_grad=True)
    src_row = torch.randperm(n_src, device=dev)[:n].contiguous()
    anchor = (torch.randn(n, 3, device=dev) * 2).requires_grad_(True)
    cam = torch.tensor([0.3, -3.0, 0.5], device=dev)
    ws = [torch.randn(n, k, device=dev) for k in (10, 30, 70)]
    params = [p for m in (mo, mc, mv) for p in m.parameters()]

    def run(tiled):
        monkeypatch.setattr(mlp, "M3_TILED", tiled)
        for t in [feat_src, anchor] + params:
            t.grad = None
        outs = mlp.anchor_mlp3_rows(feat_src, src_row, anchor, cam, mo, mc, mv)
        sum((o * w).sum() for o, w in zip(outs, ws)).backward()
        return [o.detach().clone() for o in outs] + [feat_src.grad.clone(), anchor.grad.clone()] + [p.grad.clone() for p in params]

    a, b = run(True), run(False)
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), (i, float((x - y).abs().max()))


@pytest.mark.parametrize("n", [1, 3, 17, 33, 1001])
def test_anchor_mlp3_forward_writes_nothing_behind_its_rows(n):
    """The forward stores 16-row tile images as 16-byte pieces and leaves the rows behind n to the buffer bounds check (per
    dword): outputs placed inside larger sentinel-filled buffers — nothing behind row n - 1 changes, for row counts whose last
    row ends inside a 16-byte piece (40-, 120-, 280-, 216-byte rows), on both forward entry points."""
    import torch.nn as nn
    from contextgs_amd import _lib, mlp
    torch.manual_seed(11)
    dev = "cuda"
    L = _lib.lib()
    mk = lambda out: nn.Sequential(nn.Linear(54, 50), nn.ReLU(True), nn.Linear(50, out)).to(dev)
    ms = [mk(10), mk(30), mk(70)]
    W1, b1 = [m[0].weight.detach().contiguous() for m in ms], [m[0].bias.detach().contiguous() for m in ms]
    W2, b2 = [m[2].weight.detach().contiguous() for m in ms], [m[2].bias.detach().contiguous() for m in ms]
    x = torch.randn(n, 54, device=dev)
    pad = 64
    mkbuf = lambda w: torch.full((n * w + pad,), -7.25, device=dev)
    ref = None
    for rows in (False, True):
        ys, h, xo = [mkbuf(10), mkbuf(30), mkbuf(70)], mkbuf(150), mkbuf(54)
        if rows:
            feat = x[:, :50].contiguous()
            src = torch.arange(n, device=dev)
            cam = torch.tensor([0.3, -3.0, 0.5], device=dev)
            anchor = torch.randn(n, 3, device=dev) * 2
            _lib.check(L.cgs_anchor_mlp3_forward_rows_t(_lib.ptr(feat), _lib.ptr(src), _lib.ptr(anchor), _lib.ptr(cam), _lib.ptr(xo),
                                                        mlp._ptr_array(W1), mlp._ptr_array(b1), mlp._ptr_array(W2), mlp._ptr_array(b2),
                                                        _lib.ptr(ys[0]), _lib.ptr(ys[1]), _lib.ptr(ys[2]), _lib.ptr(h), n, 0,
                                                        _lib.current_stream()), "fwd rows")
            assert bool((xo[n * 54:] == -7.25).all()) and bool((xo[:n * 54] != -7.25).all())
        else:
            _lib.check(L.cgs_anchor_mlp3_forward(_lib.ptr(x), 54, mlp._ptr_array(W1), mlp._ptr_array(b1), mlp._ptr_array(W2),
                                                 mlp._ptr_array(b2), _lib.ptr(ys[0]), _lib.ptr(ys[1]), _lib.ptr(ys[2]), _lib.ptr(h), n,
                                                 _lib.current_stream()), "fwd")
            ref = [torch.tanh(ms[0](x)).detach(), torch.sigmoid(ms[1](x)).detach(), ms[2](x).detach()]     # the heads' fixed activations
            for y, r, w in zip(ys, ref, (10, 30, 70)):
                assert torch.allclose(y[:n * w].view(n, w), r, rtol=1e-5, atol=1e-5)
        torch.cuda.synchronize()
        for y, w in zip(ys + [h], (10, 30, 70, 150)):
            assert bool((y[n * w:] == -7.25).all()), (rows, w)
            assert bool((y[:n * w] != -7.25).all()), (rows, w)


def _mlp_zoo_grads(defer, twice=False):
    """Every MLP node of the path in one graph (plain, recomputing, level node, fused anchor MLPs both ways); returns the
    gradients of all parameters and inputs with the weight gradients launched inline or at the end of the backward."""
    from contextgs_amd import mlp
    torch.manual_seed(11)
    dev = "cuda"
    grid = _seq(71, 100, 175, None)
    small = _seq(15, 100, 3, None)
    mo, mc, mv = _seq(54, 50, 10, nn.Tanh), _seq(54, 50, 30, nn.Sigmoid), _seq(54, 50, 70, None)
    n = 4099
    x71 = torch.randn(n, 71, device=dev, requires_grad=True)
    x15 = torch.randn(n, 15, device=dev, requires_grad=True)
    x54 = torch.randn(n, 54, device=dev, requires_grad=True)
    loc = torch.randperm(n, device=dev)[:700].sort().values
    feat_src = torch.randn(n + 50, 50, device=dev, requires_grad=True)
    src_row = torch.randperm(n + 50, device=dev)[:n].contiguous()
    anchor = (torch.randn(n, 3, device=dev) * 2).requires_grad_(True)
    cam = torch.tensor([0.3, -3.0, 0.5], device=dev)
    prev = mlp.defer_weight_gradients(defer)
    try:
        for _ in range(2 if twice else 1):       # the second pass ACCUMULATES into .grad
            qadj, pred = mlp.level_mlp(x71, loc, grid, 172)
            outs = [qadj, pred, mlp.mlp2(x15, small), mlp.mlp2(x71, grid), *mlp.anchor_mlp3(x54, mo, mc, mv),
                    *mlp.anchor_mlp3_rows(feat_src, src_row, anchor, cam, mo, mc, mv)]
            g = torch.Generator(device=dev).manual_seed(5)
            sum((o * torch.randn(o.shape, device=dev, generator=g)).sum() for o in outs).backward()
            assert not mlp._Deferred.queue and not mlp._Deferred.armed
    finally:
        mlp.defer_weight_gradients(prev)
    params = [p for s in (grid, small, mo, mc, mv) for p in s.parameters()]
    return [t.grad for t in params + [x71, x15, x54, feat_src, anchor]]


@pytest.mark.parametrize("twice", [False, True])
def test_deferred_weight_gradients_are_the_inline_ones(twice):
    """mlp.defer_weight_gradients (what dist.GradientSync switches on for world > 1): the data-only backward entry points +
    cgs_mlp2_wgrad / cgs_anchor_mlp3_wgrad from the autograd engine's end-of-backward callback leave bit-identical .grad on
    every parameter and input, including a second backward that accumulates.  One exception since round 4: the INLINE
    backward of the anchor MLPs forms their weight gradients inside the backward kernel (mlp3_bwd_wg_kernel: per-wave register
    accumulators, another summation order than the deferred wgrad_multi launch) — those twelve tensors (entries 8..19) agree
    to fp32 rounding, not bit for bit; each path by itself is bit-reproducible (second half of the test)."""
    a, b = _mlp_zoo_grads(False, twice), _mlp_zoo_grads(True, twice)
    assert len(a) == len(b) and all(t is not None for t in a + b)
    for i, (u, v) in enumerate(zip(a, b)):
        if 8 <= i < 20:
            assert float((u - v).abs().max()) <= 2e-5 * max(1e-6, float(v.abs().max())), i
        else:
            assert torch.equal(u, v), i
    for i, (u, v) in enumerate(zip(a, _mlp_zoo_grads(False, twice))):
        assert torch.equal(u, v), ("inline path not reproducible", i)


def test_deferral_leaves_non_leaf_weights_to_autograd_and_runs_the_hooks():
    from contextgs_amd import mlp
    torch.manual_seed(2)
    seq = _seq(71, 100, 175, None)
    x = torch.randn(300, 71, device="cuda")
    scale = torch.ones((), device="cuda", requires_grad=True)
    calls = []
    hook = mlp.add_before_flush_hook(lambda: calls.append(1))
    prev = mlp.defer_weight_gradients(True)
    try:
        # weights that are functions of another leaf: their gradient must travel through autograd, not into .grad
        y = mlp.mlp2_weights(x, seq[0].weight * scale, seq[0].bias, seq[2].weight, seq[2].bias, 0)
        y.sum().backward()
        assert scale.grad is not None and seq[2].weight.grad is not None and calls == []
        seq.zero_grad()
        mlp.mlp2(x, seq).sum().backward()                     # leaf weights: deferred, the hook runs before the launches
        assert calls == [1] and all(p.grad is not None for p in seq.parameters())
    finally:
        mlp.defer_weight_gradients(prev)
        mlp.remove_before_flush_hook(hook)


def test_data_only_backward_rejects_a_partial_pointer_set():
    from contextgs_amd import _lib
    L = _lib.lib()
    n = 64
    t = lambda *s: torch.zeros(*s, device="cuda")
    x, W1, W2, dy, h, dz1, dW2 = t(n, 71), t(100, 71), t(175, 100), t(n, 175), t(n, 100), t(n, 100), t(175, 100)
    rc = L.cgs_mlp2_backward(71, 100, 175, 0, _lib.ptr(x), 71, _lib.ptr(W1), None, _lib.ptr(W2), None, _lib.ptr(dy), 175, _lib.ptr(h),
                             None, 71, 0, _lib.ptr(dz1), None, None, None, _lib.ptr(dW2), None, n, None, 0, _lib.current_stream())
    assert rc != 0


def test_a_backward_that_dies_leaves_no_stale_deferred_jobs():
    """Deferred mode: if a backward raises after an MLP node has queued its weight-gradient job, the engine never runs the
    end-of-backward callback; the next step must neither execute that stale job nor lose its own weight gradients."""
    from contextgs_amd import mlp

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    torch.manual_seed(4)
    seq = _seq(71, 100, 175, None)
    x = torch.randn(500, 71, device="cuda")
    prev = mlp.defer_weight_gradients(True)
    try:
        x_bad = torch.randn(300, 71, device="cuda", requires_grad=True)
        y = mlp.mlp2(Boom.apply(x_bad), seq)                 # backward order: mlp node (queues its job), then Boom raises
        with pytest.raises(RuntimeError, match="boom"):
            y.sum().backward()
        assert mlp._Deferred.armed and len(mlp._Deferred.queue) == 1 and all(p.grad is None for p in seq.parameters())
        mlp.mlp2(x, seq).sum().backward()                     # a clean step afterwards
        assert not mlp._Deferred.armed and not mlp._Deferred.queue and not mlp._Deferred.staged
        got = [p.grad.clone() for p in seq.parameters()]
    finally:
        mlp.defer_weight_gradients(prev)
    seq.zero_grad()
    mlp.mlp2(x, seq).sum().backward()                         # the same step with inline weight gradients
    for a, p in zip(got, seq.parameters()):
        assert torch.equal(a, p.grad)
