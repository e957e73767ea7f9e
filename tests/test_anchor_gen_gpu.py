"""The fused anchor-MLP + expansion kernel family (csrc/anchor_gen.hip, contextgs_amd/anchor_gen.py) against
(a) a plain torch fp32 statement of gaussian_renderer/__init__.py:106-145 and (b) the unfused kernel pair it replaces.
Forward values are the same fp32 expression trees as the unfused kernels (bit-equal); gradients differ by summation
order only: 2e-5 of the tensor's max for per-row gradients, 2e-4 for the weight gradients (sums over all rows)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
K = 10


def _mlps(dev="cuda"):
    mk = lambda out, act: nn.Sequential(nn.Linear(54, 50), nn.ReLU(True), nn.Linear(50, out), *([act()] if act else [])).to(dev)
    return mk(10, nn.Tanh), mk(30, nn.Sigmoid), mk(70, None)


def _inputs(n_src, n, seed, indexed=True, dev="cuda"):
    torch.manual_seed(seed)
    feat_src = torch.randn(n_src, 50, device=dev, requires_grad=True)
    gs_src = (torch.rand(n_src, 6, device=dev) * 0.1 + 0.01).requires_grad_(True)
    off_src = (torch.randn(n_src, K, 3, device=dev) * 0.5).requires_grad_(True)
    row = torch.randperm(n_src, device=dev)[:n].contiguous() if indexed else None
    anchor = (torch.randn(n, 3, device=dev) * 2).requires_grad_(True)
    cam = torch.tensor([0.3, -3.0, 0.5], device=dev)
    masks = (torch.rand(n, K, device=dev) < 0.7).float().requires_grad_(True)
    return feat_src, gs_src, off_src, row, anchor, cam, masks


def _torch_reference(feat_src, gs_src, off_src, row, anchor, cam, masks, mo, mc, mv):
    """gaussian_renderer/__init__.py:106-145, line by line."""
    n = anchor.shape[0]
    feat = feat_src[row] if row is not None else feat_src
    gs = gs_src[row] if row is not None else gs_src
    off = off_src[row] if row is not None else off_src
    ob_view = anchor - cam
    ob_dist = ob_view.norm(dim=1, keepdim=True)
    ob_view = ob_view / ob_dist
    x = torch.cat([feat, ob_view, ob_dist], dim=1)
    neural_opacity = mo(x).reshape(-1, 1) * masks.reshape(-1, 1)
    mask = (neural_opacity > 0.0).view(-1)
    opacity = neural_opacity[mask]
    color = mc(x).reshape(n * K, 3)
    scale_rot = mv(x).reshape(n * K, 7)
    offsets = off.reshape(-1, 3)
    concatenated = torch.cat([gs, anchor], dim=-1).unsqueeze(1).expand(n, K, 9).reshape(-1, 9)
    masked = torch.cat([concatenated, color, scale_rot, offsets], dim=-1)[mask]
    scaling_repeat, repeat_anchor, color, scale_rot, offsets = masked.split([6, 3, 3, 7, 3], dim=-1)
    scaling = scaling_repeat[:, 3:] * torch.sigmoid(scale_rot[:, :3])
    rot = F.normalize(scale_rot[:, 3:7])
    xyz = repeat_anchor + offsets * scaling_repeat[:, :3]
    return xyz, color, opacity, scaling, rot, neural_opacity, mask


def _loss(outs, ws):
    return sum((o * w).sum() for o, w in zip(outs[:6], ws))


def _weights(P, n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return [torch.randn(P, k, device="cuda", generator=g) for k in (3, 3, 1, 3, 4)] + [torch.randn(n * K, 1, device="cuda", generator=g)]


def _grads(tensors):
    out = [t.grad.clone() if t.grad is not None else None for t in tensors]
    for t in tensors:
        t.grad = None
    return out


@pytest.mark.parametrize("n_src,n,indexed", [(40, 40, False), (1, 1, False), (97, 33, True), (5000, 3777, True), (60001, 60001, True)])
@pytest.mark.parametrize("fused_wgrad", [True, False])
def test_anchor_gen_matches_torch(n_src, n, indexed, fused_wgrad, monkeypatch):
    from contextgs_amd import anchor_gen
    monkeypatch.setattr(anchor_gen, "FUSED_WGRAD", fused_wgrad)
    mo, mc, mv = _mlps()
    feat_src, gs_src, off_src, row, anchor, cam, masks = _inputs(n_src, n, seed=n)
    if not indexed:
        row = None
    params = [p for m in (mo, mc, mv) for p in m.parameters()]
    leaves = [feat_src, gs_src, off_src, anchor, masks] + params
    ref = _torch_reference(feat_src, gs_src, off_src, row, anchor, cam, masks, mo, mc, mv)
    P = int(ref[0].shape[0])
    ws = _weights(P, n, seed=n + 1)
    _loss(ref, ws).backward()
    g_ref = _grads(leaves)
    got = anchor_gen.anchor_gen(feat_src, row, anchor, cam, gs_src, off_src, row, masks, mo, mc, mv)
    assert int(got[0].shape[0]) == P and torch.equal(got[6], ref[6])
    for a, b, name in zip(got[:6], ref[:6], ["xyz", "color", "opacity", "scaling", "rot", "neural_opacity"]):
        assert (a - b).abs().max() <= 2e-5 * max(1.0, float(b.detach().abs().max())), name
    _loss(got, ws).backward()
    g_got = _grads(leaves)
    names = ["feat", "gs", "off", "anchor", "masks"] + [f"p{i}" for i in range(len(params))]
    for a, b, name in zip(g_got, g_ref, names):
        assert a is not None and b is not None, name
        scale = max(1e-6, float(b.abs().max()))
        tol = (2e-4 if name.startswith("p") else 2e-5) * scale
        assert (a - b).abs().max() <= tol, (name, float((a - b).abs().max()), scale)
    if indexed and n < n_src:       # rows of the sources that no visible anchor reads get exactly zero
        unread = torch.ones(n_src, dtype=torch.bool, device="cuda")
        unread[row] = False
        for a in g_got[:3]:
            assert float(a[unread].abs().sum()) == 0.0


def test_anchor_gen_forward_is_bit_equal_to_the_unfused_kernels():
    """Same fp32 expressions and the same MFMA k order as csrc/mlp3.hip + csrc/expand.hip: identical bits."""
    from contextgs_amd import anchor_gen, mlp
    from contextgs_amd.renderer import _ExpandGaussians
    mo, mc, mv = _mlps()
    n_src, n = 30000, 20011
    feat_src, gs_src, off_src, row, anchor, cam, masks = _inputs(n_src, n, seed=7)
    with torch.no_grad():
        got = anchor_gen.anchor_gen(feat_src, row, anchor, cam, gs_src, off_src, row, masks, mo, mc, mv)
        op_raw, color_in, cov_in = mlp.anchor_mlp3_rows(feat_src, row, anchor, cam, mo, mc, mv)
        old = _ExpandGaussians.apply(anchor, gs_src, off_src, masks, op_raw, color_in, cov_in, K, row)
    for a, b in zip(got, old):
        assert torch.equal(a, b)


def test_anchor_gen_no_grad_and_empty():
    from contextgs_amd import anchor_gen
    mo, mc, mv = _mlps()
    feat_src, gs_src, off_src, row, anchor, cam, masks = _inputs(10, 0, seed=1)
    with torch.no_grad():
        out = anchor_gen.anchor_gen(feat_src, row, anchor, cam, gs_src, off_src, row, masks, mo, mc, mv)
    assert out[0].shape == (0, 3) and out[5].shape == (0, 1)
    # every offset masked away: no Gaussian survives, gradients still flow to the mask
    feat_src, gs_src, off_src, row, anchor, cam, masks = _inputs(50, 50, seed=2, indexed=False)
    zero_mask = torch.zeros_like(masks).requires_grad_(True)
    out = anchor_gen.anchor_gen(feat_src, None, anchor, cam, gs_src, off_src, None, zero_mask, mo, mc, mv)
    assert out[0].shape[0] == 0 and not bool(out[6].any())
    (out[5] * torch.arange(500, device="cuda").float().view(-1, 1)).sum().backward()
    assert zero_mask.grad is not None and float(zero_mask.grad.abs().max()) > 0


@pytest.mark.parametrize("step", [1000, 5000, 20000])
def test_render_through_fused_and_unfused_nodes_agree(step, monkeypatch):
    """render() of the three training phases with the fused node (default) and with the kernel pair it replaces:
    same image bits, every parameter gradient equal up to summation order."""
    import itertools
    from contextgs_amd import anchor_gen, ctx_ops
    from contextgs_amd.renderer import prefilter_voxel, render
    from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
    pc = make_scene(20000, seed=0)
    pc.train()
    cam = orbit_cameras(4, 320, 180)[1].to_torch("cuda")
    pipe, bg = SynthPipe(), torch.zeros(3, device="cuda")
    w = torch.randn(3, 180, 320, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    calls = []
    real = anchor_gen._AnchorGen.apply
    monkeypatch.setattr(anchor_gen._AnchorGen, "apply", staticmethod(lambda *a: (calls.append(1), real(*a))[1]))

    def run(enabled):
        monkeypatch.setattr(anchor_gen, "ENABLED", enabled)
        monkeypatch.setattr(anchor_gen, "MODE", "all" if enabled else "off")
        torch.manual_seed(11)                     # phase 2 draws torch noise
        monkeypatch.setattr(ctx_ops, "_seed_counter", itertools.count(1))   # same counter-based noise streams in both runs
        leaves = [pc._anchor, pc._offset, pc._mask, pc._anchor_feat, pc._scaling] + \
            [p for m in (pc.mlp_opacity, pc.mlp_color, pc.mlp_cov) for p in m.parameters()]
        for t in leaves:
            t.grad = None
        vis = prefilter_voxel(cam, pc, pipe, bg)
        pkg = render(cam, pc, pipe, bg, visible_mask=vis, retain_grad=True, step=step)
        loss = (pkg["render"] * w).sum() + 0.01 * pkg["scaling"].prod(dim=1).mean()
        if step > 10000:
            loss = loss + 0.001 * pkg["bit_per_param"]
        loss.backward()
        return pkg["render"].detach().clone(), [t.grad.clone() for t in leaves], pkg["viewspace_points"].grad.clone()

    img_f, g_f, vs_f = run(True)
    assert calls, "the fused node was not used"
    n_calls = len(calls)
    img_u, g_u, vs_u = run(False)
    assert len(calls) == n_calls
    if step <= 3000 or step > 10000:              # phase 2 adds torch noise per call: same seed, same draws
        assert torch.equal(img_f, img_u)
    assert (img_f - img_u).abs().max() <= 1e-6
    for a, b in zip(g_f + [vs_f], g_u + [vs_u]):
        scale = max(1e-6, float(b.abs().max()))
        assert (a - b).abs().max() <= 3e-4 * scale, (tuple(a.shape), float((a - b).abs().max()), scale)
