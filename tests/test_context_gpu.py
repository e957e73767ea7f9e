"""GPU parity of the Python half (quantisers, rate model, level division, context model,
expansion) against the REFERENCE's outputs in tests/golden/*.npz, through the drop-in
modules of contextgs_amd (which call libcgs_hip.so via the C-ABI).

Bit-exact: Quantize_anchor, STE_*, level indices.  Floating point: tolerances stated inline.
"""
import os
import types

import numpy as np
import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def _sub(g, a):
    """Rows of a full-size array that a (possibly row-strided, tools/make_goldens.pack) fixture holds."""
    return a[::int(g["stride"])] if "stride" in g.files else a


def _colsums_ok(g, key, a, rtol):
    """All rows of a strided fixture through the fp64 column sums it carries."""
    if "stride" not in g.files:
        return True
    a2 = a.reshape(a.shape[0], -1).astype(np.float64)
    return bool(np.all(np.abs(a2.sum(0) - g[key + "__colsum"]) <= rtol * g[key + "__abssum"] + 1e-9))


def _align(a, b, tol, max_skips):
    """Index pairs (i, j) of rows a[i] ~ b[j] of two compacted row lists that differ by a few inserted / dropped rows
    (two-pointer walk on row equality within tol)."""
    i = j = skips = 0
    pairs = []
    while i < a.shape[0] and j < b.shape[0]:
        if np.all(np.abs(a[i] - b[j]) <= tol * (1 + np.abs(b[j]))):
            pairs.append((i, j)); i += 1; j += 1
            continue
        skips += 1
        assert skips <= max_skips, "too many unmatched rows"
        # which side holds the extra row?  the one whose NEXT row matches the other's current row
        if i + 1 < a.shape[0] and np.all(np.abs(a[i + 1] - b[j]) <= tol * (1 + np.abs(b[j]))):
            i += 1
        elif j + 1 < b.shape[0] and np.all(np.abs(a[i] - b[j + 1]) <= tol * (1 + np.abs(b[j + 1]))):
            j += 1
        else:                # a substituted row (its values moved by a quantisation step): skip both
            i += 1; j += 1
    return np.array(pairs, dtype=np.int64).reshape(-1, 2)


MODEL_CASES = [("n64", 64, 1), ("n3000", 3000, 2), ("n10000", 10000, 4)]


def _model(N, seed):
    from contextgs_amd.model import GaussianModel
    pc = GaussianModel(feat_dim=gi.D, n_offsets=gi.K, voxel_size=0.01, level_num=gi.LEVELS, target_ratio=0.2)
    sd = pc.state_dict()
    for k, v in gi.mlp_weights(seed).items():
        assert k in sd, k
        sd[k] = T(v)
    pc.load_state_dict(sd, strict=False)
    st = gi.anchor_state(N, seed)
    pc.set_state(st["anchor"], st["offset"], st["mask"], st["feat"], st["hyper"], st["scaling"])
    pc.update_anchor_bound()
    return pc, st


def test_quantisers_bit_exact():
    from contextgs_amd.encodings import Quantize_anchor, STE_binary, STE_multistep
    g = _load("elementwise.npz")
    st = gi.anchor_state(1000, 3)
    aq, q = Quantize_anchor.apply(T(st["anchor"]), T(g["qa_min"]), T(g["qa_max"]))
    assert np.array_equal(q.cpu().numpy(), g["qa_quantized"])
    assert np.array_equal(aq.cpu().numpy(), g["qa_anchor_q"])
    x, mean, scale, Q = gi.elementwise_inputs(257, 1)
    assert np.array_equal(STE_multistep.apply(T(x), T(Q)).cpu().numpy(), g["ste_rowQ"])
    assert np.array_equal(STE_multistep.apply(T(x), T(np.broadcast_to(Q, x.shape))).cpu().numpy(), g["ste_elemQ"])
    off = x[:, :30].reshape(-1, 10, 3)
    assert np.array_equal(STE_multistep.apply(T(off), T(Q).unsqueeze(1)).cpu().numpy(), g["ste_offsets"])
    assert np.array_equal(STE_binary.apply(T(x / 3)).cpu().numpy(), g["ste_binary"])
    # gradients are straight-through
    xa = T(x).requires_grad_(True)
    STE_multistep.apply(xa, T(Q)).sum().backward()
    assert torch.equal(xa.grad, torch.ones_like(xa))


def test_entropy_gaussian_matches_reference():
    from contextgs_amd.entropy_models import Entropy_bernoulli, Entropy_gaussian
    g = _load("elementwise.npz")
    x, mean, scale, Q = gi.elementwise_inputs(257, 1)
    xg, mg, sg, Qg = (T(v).requires_grad_(True) for v in (x, mean, scale, Q))
    bits = Entropy_gaussian(Q=1)(xg, mg, sg, Qg, torch.tensor(0.25, device="cuda"))
    (bits * T(g["eg_gw"])).sum().backward()
    b = bits.detach().cpu().numpy()
    # difference of two fp32 CDFs: absolute error a few 1e-8 in the likelihood whatever erf is used
    assert np.abs(np.exp2(-b) - np.exp2(-g["eg_bits"])).max() <= 3e-7
    assert np.abs(b - g["eg_bits"]).max() <= 0.1
    well = g["eg_bits"] < 10
    for a, ref in ((xg.grad, g["eg_gx"]), (mg.grad, g["eg_gmean"]), (sg.grad, g["eg_gscale"])):
        a = a.cpu().numpy()
        assert np.allclose(a[well], ref[well], rtol=2e-3, atol=1e-5 * np.abs(ref).max())
        assert np.allclose(a[~well], ref[~well], rtol=0.15, atol=1e-3 * np.abs(ref).max())
    assert np.allclose(Qg.grad.cpu().numpy(), g["eg_gQ"], rtol=0.05, atol=1e-2 * np.abs(g["eg_gQ"]).max())
    b2 = Entropy_gaussian(Q=1)(T(x), T(mean), T(scale), T(Q)).cpu().numpy()
    assert np.abs(np.exp2(-b2) - np.exp2(-g["eg_bits_defaultmean"])).max() <= 3e-7
    b3 = Entropy_gaussian(Q=0.5)(T(x), T(mean), T(scale)).cpu().numpy()
    assert np.abs(np.exp2(-b3) - np.exp2(-g["eg_bits_scalarQ"])).max() <= 3e-7
    eb = Entropy_bernoulli()(torch.tensor([1., -1.], device="cuda"), torch.tensor([.7, .7], device="cuda"))
    assert np.allclose(eb.cpu().numpy(), g["eb_bits"], atol=1e-6)


@pytest.mark.parametrize("tag,N,seed", MODEL_CASES)
def test_levels_and_context_model(tag, N, seed):
    from contextgs_amd import context_model as cm
    from contextgs_amd.multi_level import torch_unique_with_indices
    g = _load(f"model_{tag}.npz")
    pc, st = _model(N, seed)
    with torch.no_grad():
        assert np.array_equal(pc.get_mask.cpu().numpy(), g["get_mask"])
        assert np.array_equal(pc.get_mask_anchor.cpu().numpy(), g["get_mask_anchor"])
        assert np.array_equal(pc.x_bound_min.cpu().numpy(), g["x_bound_min"])
        assert np.array_equal(pc.get_anchor.cpu().numpy(), g["get_anchor"])
        assert np.allclose(pc.get_scaling.cpu().numpy(), g["get_scaling"], rtol=2e-6)
        anchor = pc.get_anchor
        mab = pc.get_mask_anchor
        ls = cm.find_divide_scale(pc, anchor[mab], pc.target_ratio, pc.level_num)
        assert np.allclose(ls, g["level_scale"], rtol=1e-6)
        pc.level_scale = [float(v) for v in g["level_scale"]]
        key = torch.round(anchor / pc.voxel_size / pc.level_scale[0])
        u, inv, idx, cnt = torch_unique_with_indices(key, dim=0)
        assert np.array_equal(u.cpu().numpy(), g["uniq_rows"]) and np.array_equal(inv.cpu().numpy(), g["uniq_inverse"])
        assert np.array_equal(idx.cpu().numpy(), g["uniq_indices"]) and np.array_equal(cnt.cpu().numpy(), g["uniq_counts"])
        for variant, src, m in (("train", anchor, mab), ("enc", anchor[mab], None)):
            _h, il, ml, last = cm.divide_levels(pc, src, m)
            for i in range(2):
                assert np.array_equal(il[i].cpu().numpy(), g[f"div_{variant}_inverse{i}"])
                assert np.array_equal(ml[i].cpu().numpy(), g[f"div_{variant}_mapping{i}"])
            assert np.array_equal(last.cpu().numpy(), g[f"div_{variant}_last"])
        pc.eval()
        # use the reference's exp(scaling) so that only the context model is under test
        f, s, o = cm.multi_scale_generating(pc, anchor, pc._hyper_latent, pc._anchor_feat, pc._offset,
                                            T(g["get_scaling"]), pc.get_mask, mab, predict_bpp=False, training=False)
        for a, b, step in ((f, g["msg_feat"], 1.0), (s, g["msg_scaling"], 1e-3), (o, g["msg_offsets"], 0.2)):
            d = np.abs(_sub(g, a.cpu().numpy()) - b)
            bad = d > 1e-4 * step
            print(f"[allowance] msg eval {tag} step {step:g}: {int(bad.sum())} of {bad.size} entries off by a quantisation step "
                  f"(allowed {int(5e-4 * bad.size)}), max |diff| {float(d.max()):.3g} (allowed {2.02 * step:.3g})")
            assert bad.mean() <= 5e-4, bad.mean()          # rounding-boundary flips only ...
            assert d.max() <= 2.02 * step                  # ... and a flip moves a value by ONE step Q < 2 Q0
        sums = cm.multi_scale_generating(pc, anchor[mab], pc._hyper_latent[mab], pc._anchor_feat[mab], pc._offset[mab],
                                         T(g["get_scaling"])[mab], binary_grid_masks=pc.get_mask[mab], predict_bpp=True,
                                         return_sum_bits=True)
        assert sums[0] == g["msg_sum_bits"][0]
        assert np.allclose(sums[1:], g["msg_sum_bits"][1:], rtol=2e-3)


@pytest.mark.parametrize("tag,N,seed", MODEL_CASES)
def test_generate_neural_gaussians(tag, N, seed):
    from contextgs_amd.renderer import generate_neural_gaussians
    g = _load(f"model_{tag}.npz")
    pc, st = _model(N, seed)
    pc.level_scale = [float(v) for v in g["level_scale"]]
    cam = types.SimpleNamespace(camera_center=T(gi.camera_center(seed)))
    vis = T(g["visible_mask"])
    close = lambda a, b: a.shape == b.shape and np.allclose(a, b, rtol=1e-4, atol=3e-6)
    with torch.no_grad():
        pc.eval()
        xyz, color, opacity, scaling, rot, _ = generate_neural_gaussians(cam, pc, vis, is_training=False)
    n_ref, n_got = int(g["ev_count"]), xyz.shape[0]
    outs = [(xyz, "ev_xyz"), (color, "ev_color"), (opacity, "ev_opacity"), (scaling, "ev_scaling"), (rot, "ev_rot")]
    if n_got == n_ref:
        for a, key in outs:
            b = g[key]
            d = np.abs(_sub(g, a.cpu().numpy()) - b)
            out_ = d > 1e-4 * (1 + np.abs(b))
            print(f"[allowance] expansion eval {tag} {key}: {int(out_.sum())} of {out_.size} outside 1e-4 (allowed {int(2e-3 * out_.size)})")
            assert out_.mean() <= 2e-3, key
    else:
        # a rounding-boundary flip in the context model moved a borderline opacity across 0: a Gaussian appears or
        # disappears.  At most a handful, and every OTHER row must still match: align the two row lists on xyz
        print(f"[allowance] expansion eval {tag}: {n_got} Gaussians vs {n_ref} in the fixture (allowed difference {max(2, n_ref // 2000)}); "
              "rows aligned on xyz")
        assert abs(n_got - n_ref) <= max(2, n_ref // 2000)
        assert "stride" not in g.files, "row alignment needs the full fixture"
        pairs = _align(xyz.cpu().numpy(), g["ev_xyz"], 1e-4, max(4, n_ref // 1000))
        assert pairs.shape[0] >= n_ref - max(4, n_ref // 1000)
        for a, key in outs[1:]:
            a_, b_ = a.cpu().numpy()[pairs[:, 0]], g[key][pairs[:, 1]]
            out_ = np.abs(a_ - b_) > 1e-4 * (1 + np.abs(b_))
            print(f"[allowance] expansion eval {tag} {key} (aligned rows): {int(out_.sum())} of {out_.size} outside 1e-4 "
                  f"(allowed {int(2e-3 * out_.size)})")
            assert out_.mean() <= 2e-3, key

    pc.train()
    res = generate_neural_gaussians(cam, pc, vis, is_training=True, step=1000)
    xyz, color, opacity, scaling, rot, neural_opacity, mask = res[:7]
    assert res[7] is None and res[8] == 16
    assert np.array_equal(mask.cpu().numpy(), g["tr_mask"])
    for a, key in ((xyz, "tr_xyz"), (color, "tr_color"), (opacity, "tr_opacity"), (scaling, "tr_scaling"),
                   (rot, "tr_rot"), (neural_opacity, "tr_neural_opacity")):
        a = a.detach().cpu().numpy()
        assert close(_sub(g, a), g[key]) and _colsums_ok(g, key, a, 1e-4), key
    rng = np.random.default_rng(seed + 11)
    ws = [T(rng.normal(size=tuple(t.shape)).astype(np.float32)) for t in (xyz, color, opacity, scaling, rot)]
    loss = sum((t * w).sum() for t, w in zip((xyz, color, opacity, scaling, rot), ws))
    loss.backward()
    assert abs(loss.item() - float(g["tr_loss"])) <= 1e-4 * abs(float(g["tr_loss"])) + 1e-3
    for got, key in ((pc._anchor.grad, "g_anchor"), (pc._offset.grad, "g_offset"), (pc._mask.grad, "g_mask"),
                     (pc._anchor_feat.grad, "g_feat"), (pc._scaling.grad, "g_scaling"),
                     (pc.mlp_opacity[2].weight.grad, "g_op_w2"), (pc.mlp_cov[0].weight.grad, "g_cov_w0"),
                     (pc.mlp_color[2].bias.grad, "g_color_b2")):
        ref = g[key]
        a = got.cpu().numpy()
        if key in ("g_anchor", "g_offset", "g_mask", "g_feat", "g_scaling"):
            assert _colsums_ok(g, key, a, 1e-3), key
            a = _sub(g, a)
        assert a.shape == ref.shape, key
        assert np.abs(a - ref).max() <= 2e-4 * max(1e-6, np.abs(ref).max()), (key, np.abs(a - ref).max(), np.abs(ref).max())


def test_extract_context_feat_matches_reference_directly():
    """Row b3 on its own (scene/gaussian_model.py:1711-1724): the reference's extract_context_feat called on the n3000
    model's level division with two `already_coded` patterns per level (tests/golden/context_feat.npz) — same rows in the
    same order (quirk Q1: ascending original index), bit-equal values."""
    from contextgs_amd import context_model as cm
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "context_feat.npz"))
    pc, _st = _model(3000, 2)
    pc.level_scale = [float(v) for v in z["level_scale"]]
    T = lambda k: torch.tensor(z[k], device="cuda")
    anchor, feat, scal = T("anchor_q"), T("feat"), T("scaling")
    assert torch.equal(pc.get_anchor.detach(), anchor)
    _, inverses, mappings, _ = cm.divide_levels(pc, anchor, torch.ones(3000, dtype=torch.bool, device="cuda"))
    n_cases = 0
    for key in z.files:
        if not key.startswith("ctx_l"):
            continue
        level = int(key[5])
        coded = T("coded" + key[3:])
        got = cm.extract_context_feat(anchor, feat, scal, coded, inverses, mappings, level)
        assert got.shape == z[key].shape and torch.equal(got, T(key)), key
        n_cases += 1
    assert n_cases == 4


@pytest.mark.parametrize("n,span", [(1, 3), (5000, 7), (200_003, 90), (60_000, 1 << 18), (40_000, 1 << 12)])
def test_level_unique_kernels_equal_torch_unique(n, span):
    """csrc/levels.hip (cgs_level_key_range + cgs_level_unique) against torch.unique(dim=0) on the host, which is what
    utils/multi_level.py:3-31 calls: lexicographic row order, inverse, smallest original index per group, counts — for
    narrow keys (one 32-bit sort) and wide ones (> 32 packed bits: two sorts), negative coordinates and -0.0."""
    from contextgs_amd.multi_level import torch_unique_with_indices
    g = torch.Generator().manual_seed(n + span)
    keys = torch.randint(-span, span + 1, (n, 3), generator=g).float()
    keys[::7, 1] = -0.0 * keys[::7, 1].abs().clamp(max=0)          # a few explicit -0.0
    u, inv, first, cnt = torch_unique_with_indices(keys.cuda(), dim=0)
    ru, rinv, rcnt = torch.unique(keys + 0.0, dim=0, return_inverse=True, return_counts=True)
    rfirst = torch.full((ru.shape[0],), n, dtype=torch.long).scatter_reduce_(0, rinv, torch.arange(n), reduce="amin")
    assert torch.equal(u.cpu(), ru) and torch.equal(inv.cpu(), rinv) and torch.equal(cnt.cpu(), rcnt)
    assert torch.equal(first.cpu(), rfirst)
    assert not torch.signbit(u).logical_and(u == 0).any()           # -0.0 merged into +0.0


def test_level_unique_rejects_non_integer_rows():
    from contextgs_amd.multi_level import torch_unique_with_indices
    with pytest.raises(NotImplementedError):
        torch_unique_with_indices(torch.tensor([[0.5, 1.0, 2.0]], device="cuda"), dim=0)
