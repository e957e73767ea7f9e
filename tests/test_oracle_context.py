"""The numpy oracle of the Python half (oracle/context_ref.py) against the fixtures
produced by the REFERENCE's own code (tests/golden/*.npz, tools/make_goldens.py)."""
import os

import numpy as np
import pytest

import golden_inputs as gi
from oracle import context_ref as cr

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def _sub(g, a):
    """Rows of a full-size array that a (possibly row-strided, tools/make_goldens.pack) fixture holds."""
    return a[::int(g["stride"])] if "stride" in g.files else a


MODEL_CASES = [("n64", 64, 1), ("n3000", 3000, 2), ("n10000", 10000, 4)]


def test_quantize_anchor_bit_exact():
    g = _load("elementwise.npz")
    st = gi.anchor_state(1000, 3)
    aq, q = cr.quantize_anchor(st["anchor"], g["qa_min"], g["qa_max"])
    assert np.array_equal(q, g["qa_quantized"])
    assert np.array_equal(aq, g["qa_anchor_q"])
    aq1, q1 = cr.quantize_anchor([[.1, .2, .3]], [[-1, -1, -1]], [[1, 1, 1]])
    assert np.array_equal(q1, g["qa_small_q"]) and q1.tolist() == [[34900, 38073, 41246]]


def test_ste_bit_exact():
    g = _load("elementwise.npz")
    x, mean, scale, Q = gi.elementwise_inputs(257, 1)
    assert np.array_equal(cr.ste_multistep(x, Q), g["ste_rowQ"])
    assert np.array_equal(cr.ste_multistep(x, np.broadcast_to(Q, x.shape)), g["ste_elemQ"])
    off = x[:, :30].reshape(-1, 10, 3)
    assert np.array_equal(cr.ste_multistep(off, Q[:, None, :]), g["ste_offsets"])
    assert np.array_equal(cr.ste_binary(x / 3), g["ste_binary"])


def test_entropy_gaussian_forward_and_grads():
    g = _load("elementwise.npz")
    x, mean, scale, Q = gi.elementwise_inputs(257, 1)
    # bits = -log2(Phi(u) - Phi(l)): the difference of two fp32 CDFs carries an absolute error of a few
    # 1e-8 whatever erf is used, so compare likelihoods (2^-bits) absolutely and bits loosely
    def close(a, b):
        return np.abs(np.exp2(-a) - np.exp2(-b)).max() <= 3e-7 and np.abs(a - b).max() <= 0.1
    bits = cr.entropy_gaussian(x, mean, scale, Q, 0.25)
    assert close(bits, g["eg_bits"])
    assert close(cr.entropy_gaussian(x, mean, scale, Q), g["eg_bits_defaultmean"])
    assert close(cr.entropy_gaussian(x, mean, scale, np.float32(0.5)), g["eg_bits_scalarQ"])
    gx, gm, gs, gq = cr.entropy_gaussian_grads(x, mean, scale, Q, 0.25, g["eg_gw"])
    for a, b in ((gx, g["eg_gx"]), (gm, g["eg_gmean"]), (gs, g["eg_gscale"]), (gq.sum(1, keepdims=True), g["eg_gQ"])):
        # the gradient carries 1/likelihood: where the likelihood is tiny its fp32 cancellation error (a few
        # 1e-8 absolute) dominates, so: tight where likelihood > 1e-3, 15 % elsewhere
        if a.shape == bits.shape:
            well = g["eg_bits"] < 10
            assert np.allclose(a[well], b[well], rtol=2e-3, atol=1e-5 * np.abs(b).max())
            assert np.allclose(a[~well], b[~well], rtol=0.15, atol=1e-3 * np.abs(b).max())
        else:
            assert np.allclose(a, b, rtol=0.05, atol=1e-2 * np.abs(b).max())   # row sum over 50 such terms
    assert np.allclose(cr.entropy_bernoulli([1, -1], [.7, .7]), g["eb_bits"], atol=1e-6)
    assert np.allclose(g["eb_bits"], [0.5146, 1.7370], atol=1e-4)


@pytest.mark.parametrize("tag,N,seed", MODEL_CASES)
def test_levels_and_context_model(tag, N, seed):
    g = _load(f"model_{tag}.npz")
    W = gi.mlp_weights(seed)
    st = gi.anchor_state(N, seed)
    # accessors (scene/gaussian_model.py:288-345)
    s = (1 / (1 + np.exp(-st["mask"].astype(np.float32)))).astype(np.float32)
    mask = (s > 0.01).astype(np.float32)
    assert np.allclose(g["get_mask"], mask, atol=1e-6)
    mab = mask.sum(1)[:, 0] > 0
    assert np.array_equal(mab, g["get_mask_anchor"])
    lo = st["anchor"].min(0, keepdims=True)
    hi = st["anchor"].max(0, keepdims=True)
    lo = np.where(lo < 0, lo * np.float32(1.2), lo * np.float32(0.8)).astype(np.float32)
    hi = np.where(hi > 0, hi * np.float32(1.2), hi * np.float32(0.8)).astype(np.float32)
    assert np.array_equal(lo, g["x_bound_min"]) and np.array_equal(hi, g["x_bound_max"])
    anchor, _q = cr.quantize_anchor(st["anchor"], lo, hi)
    assert np.array_equal(anchor, g["get_anchor"])
    scaling = np.exp(st["scaling"]).astype(np.float32)
    assert np.allclose(scaling, g["get_scaling"], rtol=1e-6)

    ls = cr.find_divide_scale(anchor[mab], lo, hi, 0.01, 0.2, 3)
    assert np.allclose(ls, g["level_scale"], rtol=1e-6)
    ls = [float(v) for v in g["level_scale"]]
    u, inv, first, cnt = cr.unique_with_indices(cr.voxel_key(anchor, 0.01, ls[0]))
    assert np.array_equal(u, g["uniq_rows"]) and np.array_equal(inv, g["uniq_inverse"])
    assert np.array_equal(first, g["uniq_indices"]) and np.array_equal(cnt, g["uniq_counts"])
    for variant, src, m in (("train", anchor, mab), ("enc", anchor[mab], None)):
        _a, il, ml, last = cr.divide_levels(src, 0.01, ls, 3, m)
        for i in range(2):
            assert np.array_equal(il[i], g[f"div_{variant}_inverse{i}"])
            assert np.array_equal(ml[i], g[f"div_{variant}_mapping{i}"])
        assert np.array_equal(last, g[f"div_{variant}_last"])

    fq, sq, oq = cr.multi_scale_generating(W, anchor, st["hyper"], st["feat"], st["offset"], g["get_scaling"], mask,
                                           mab, 0.01, ls)
    # quantised outputs are multiples of a predicted step: a 1-ulp difference in the MLP can move a value
    # sitting exactly on a rounding boundary by one step; allow a 5e-4 fraction of such flips
    for a, b, step in ((fq, g["msg_feat"], 1.0), (sq, g["msg_scaling"], 1e-3), (oq, g["msg_offsets"], 0.2)):
        d = np.abs(_sub(g, a) - b)
        bad = d > 1e-4 * step
        assert bad.mean() <= 5e-4, bad.mean()
        assert d.max() <= 2.02 * step           # a flip moves a value by ONE step, and Q = Q0 (1 + tanh) < 2 Q0
    sums = cr.multi_scale_generating(W, anchor[mab], st["hyper"][mab], st["feat"][mab], st["offset"][mab],
                                     g["get_scaling"][mab], mask[mab], None, 0.01, ls, return_sum_bits=True,
                                     x_means=(st["feat"].mean(dtype=np.float32), g["get_scaling"].mean(dtype=np.float32),
                                              st["offset"].mean(dtype=np.float32)))
    assert sums[0] == g["msg_sum_bits"][0]
    assert np.allclose(sums[1:], g["msg_sum_bits"][1:], rtol=2e-3)


@pytest.mark.parametrize("tag,N,seed", MODEL_CASES)
def test_expansion_forward(tag, N, seed):
    g = _load(f"model_{tag}.npz")
    W = gi.mlp_weights(seed)
    st = gi.anchor_state(N, seed)
    vis = g["visible_mask"]
    cam = gi.camera_center(seed)
    mask = g["get_mask"]
    # training phase step<=3000: raw parameters of the visible anchors (gaussian_renderer/__init__.py:44-50)
    xyz, color, op, sc, rot, no, sel = cr.expand(W, g["get_anchor"][vis], st["feat"][vis], st["offset"][vis],
                                                 g["get_scaling"][vis], mask[vis], cam)
    assert np.array_equal(sel, g["tr_mask"])
    for a, b in ((xyz, g["tr_xyz"]), (color, g["tr_color"]), (op, g["tr_opacity"]), (sc, g["tr_scaling"]),
                 (rot, g["tr_rot"]), (no, g["tr_neural_opacity"])):
        a = _sub(g, a)
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-4, atol=2e-6)
    # eval over the context model (:83-101)
    if "stride" in g.files:         # the large fixture keeps a row subset of the context model's outputs: take the oracle's
        mab = mask.sum(1)[:, 0] > 0
        fq, sq, oq = cr.multi_scale_generating(W, g["get_anchor"], st["hyper"], st["feat"], st["offset"], g["get_scaling"],
                                               mask, mab, 0.01, [float(v) for v in g["level_scale"]])
    else:
        fq, sq, oq = g["msg_feat"], g["msg_scaling"], g["msg_offsets"]
    xyz, color, op, sc, rot, _no, _sel = cr.expand(W, g["get_anchor"][vis], fq[vis], oq[vis], sq[vis], mask[vis], cam)
    if xyz.shape[0] != int(g["ev_count"]):      # a rounding-boundary flip in the context model moved an opacity across 0
        assert abs(xyz.shape[0] - int(g["ev_count"])) <= 2
        return
    for a, b in ((xyz, g["ev_xyz"]), (color, g["ev_color"]), (op, g["ev_opacity"]), (sc, g["ev_scaling"]), (rot, g["ev_rot"])):
        a = _sub(g, a)
        assert a.shape == b.shape and (np.abs(a - b) > 1e-4 * (1 + np.abs(b))).mean() <= 2e-3


def _strided(g, key, a):
    """Compare helper for fixtures written by tools/make_goldens.pack: (rows the fixture holds, the same rows of a)."""
    stride = int(g["stride"]) if "stride" in g.files else 1
    return g[key], a.reshape(a.shape[0], -1)[::stride].reshape(g[key].shape)


@pytest.mark.parametrize("tag,N,seed", [("n3000", 3000, 2), ("n10000", 10000, 4)])
def test_training_variant_forward(tag, N, seed):
    """Oracle of the TRAINING context / rate path (scene/gaussian_model.py:1594-1707, training=True,
    predict_bpp=True) against the reference run with the same fixed noise (tests/golden/train_*.npz)."""
    g = _load(f"train_{tag}.npz")
    W = gi.mlp_weights(seed, positive_scales=True)
    st = gi.anchor_state(N, seed)
    s = (1 / (1 + np.exp(-st["mask"].astype(np.float32)))).astype(np.float32)
    mask = (s > 0.01).astype(np.float32)
    mab = mask.sum(1)[:, 0] > 0
    lo = st["anchor"].min(0, keepdims=True)
    hi = st["anchor"].max(0, keepdims=True)
    lo = np.where(lo < 0, lo * np.float32(1.2), lo * np.float32(0.8)).astype(np.float32)
    hi = np.where(hi > 0, hi * np.float32(1.2), hi * np.float32(0.8)).astype(np.float32)
    anchor, _q = cr.quantize_anchor(st["anchor"], lo, hi)
    scaling = np.exp(st["scaling"]).astype(np.float32)
    ls = [float(v) for v in g["level_scale"]]
    train = dict(seeds=[int(v) for v in g["level_seeds"]], hyper_seed=int(g["hyper_seed"]), choose_mask=g["choose_mask"])
    fq, sq, oq, rates, extra = cr.multi_scale_generating(
        W, anchor, st["hyper"], st["feat"], st["offset"], scaling, mask, mab, 0.01, ls, train=train,
        x_means=(st["feat"].mean(dtype=np.float32), scaling.mean(dtype=np.float32), st["offset"].mean(dtype=np.float32)))
    # noisy "quantised" tensors: x + u Q with Q from the level MLP -> fp32 round-off of the MLP only
    for key, a, step in (("msg_feat", fq, 1.0), ("msg_scaling", sq, 1e-3), ("msg_offsets", oq.reshape(N, -1), 0.2)):
        ref, got = _strided(g, key, a)
        assert np.abs(got - ref).max() <= 2e-5 * step * (1 + np.abs(ref).max()), key
    # the level MLP outputs themselves (mu, sigma, step-size logits), rows strided by max(stride, 4)
    lvl_stride = max(int(g["stride"]), 4)
    for j, i in enumerate(reversed(range(3))):
        ref = g[f"pred_level{i}"]
        got = extra["preds"][j][::lvl_stride]
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-5 * (1 + np.abs(ref).max())
    got = [rates["bit_per_param"], rates["bit_per_feat_param"], rates["bit_per_scaling_param"], rates["bit_per_offsets_param"]]
    assert np.allclose(got, g["bits"], rtol=2e-4), (got, g["bits"])
    assert np.allclose(rates["each_level_bpp"][:2], g["bpp_head"], rtol=2e-4)
    assert np.allclose(np.array(rates["each_level_bpp"][2:]), g["bpp_levels"], rtol=2e-4)


def test_ctx_noise_is_uniform_and_keyed():
    """The build's counter-based noise (restated in the oracle, fed to the reference by make_goldens): range,
    moments, and independence of (seed, tensor)."""
    u = cr.ctx_noise(12345, 0, 200000)
    assert u.dtype == np.float32 and u.min() >= -0.5 and u.max() < 0.5
    assert abs(float(u.mean())) < 3e-3 and abs(float(u.var()) - 1 / 12) < 1e-3
    v = cr.ctx_noise(12345, 1, 200000)
    w = cr.ctx_noise(12346, 0, 200000)
    assert abs(float(np.corrcoef(u, v)[0, 1])) < 0.01 and abs(float(np.corrcoef(u, w)[0, 1])) < 0.01
    assert abs(float(np.corrcoef(u[:-1], u[1:])[0, 1])) < 0.01


def test_entropy_api_goldens():
    """b5 / b8 / b10 (dead-but-public API + the factorised-prior density): oracle vs the reference's outputs."""
    g = _load("entropy_api.npz")
    x, mean, scale, Q = gi.elementwise_inputs(193, 6)
    close = lambda a, b: np.abs(np.exp2(-a) - np.exp2(-b)).max() <= 3e-7 and np.abs(a - b).max() <= 0.1
    assert close(cr.entropy_gaussian(x, mean, scale, Q), g["egc_bits"])                    # Entropy_gaussian_clamp
    assert close(cr.entropy_gaussian(x, mean, scale, np.float32(0.25)), g["egc_bits_scalarQ"])
    gx, gm, gs, gq = cr.entropy_gaussian_grads(x, mean, scale, Q, x.mean(dtype=np.float32), g["egc_gw"])
    well = g["egc_bits"] < 10
    for a, b in ((gx, g["egc_gx"]), (gm, g["egc_gmean"]), (gs, g["egc_gscale"])):
        assert np.allclose(a[well], b[well], rtol=2e-3, atol=1e-5 * np.abs(b).max())
    rng = np.random.default_rng(21)
    for k, (n, p1) in enumerate(((1000, 0.7), (30, 0.0), (30, 1.0), (77777, 0.013))):
        m = (rng.random((n, 10, 1)) < p1).astype(np.float32)
        got = cr.binary_vxl_size(m)
        assert np.allclose(got, g[f"bvs_{k}"][:4], rtol=1e-6), (k, got, g[f"bvs_{k}"])
    # UniverseQuant: subtractive dither -> error uniform on [-1/2, 1/2]
    assert abs(g["uq_err_mean"]) < 3e-3 and abs(g["uq_err_var"] - 1 / 12) < 1e-3 and g["uq_err_absmax"] <= 0.5
    assert bool(g["uq_grad_is_one"])
    for seed in (2, 7):
        W = gi.mlp_weights(seed)
        v = gi.factorized_inputs(seed)
        x3 = v.T.reshape(gi.H, 1, -1)
        back = lambda t: t.reshape(gi.H, -1).T
        lower, upper = back(cr.bottleneck_logits(x3 - np.float32(0.5), W)), back(cr.bottleneck_logits(x3 + np.float32(0.5), W))
        assert np.abs(lower - g[f"fz{seed}_lower"]).max() <= 2e-5 * (1 + np.abs(g[f"fz{seed}_lower"]).max())
        assert np.abs(upper - g[f"fz{seed}_upper"]).max() <= 2e-5 * (1 + np.abs(g[f"fz{seed}_upper"]).max())
        lik = cr.bottleneck_likelihood(v, W)
        assert np.abs(lik - g[f"fz{seed}_lik"]).max() <= 3e-7
        assert np.abs(-np.log2(np.maximum(lik, 1e-6)) - g[f"fz{seed}_bits"]).max() <= 1e-3


def test_host_density_classes_match_reference(host_density):
    """The torch statement of the factorised density (entropy_bottleneck.cumulative_logits, shared by
    EntropyBottleneck and utils.entropy_models.Entropy_factorized) against the reference's `_logits_cumulative`
    outputs and autograd gradients — on CPU tensors: this is host code, no kernel involved."""
    import torch
    from contextgs_amd.entropy_bottleneck import EntropyBottleneck
    from contextgs_amd.entropy_models import Entropy_factorized
    g = _load("entropy_api.npz")
    for seed in (2, 7):
        W = gi.mlp_weights(seed)
        m = Entropy_factorized(channel=gi.H, filters=(3, 3, 3, 3))
        eb = EntropyBottleneck(gi.H)
        with torch.no_grad():
            for i in range(5):
                for dst in (m._matrices[i], eb.matrices[i]):
                    dst.copy_(torch.from_numpy(W[f"latent_codec.matrices.{i}"]))
                for dst in (m._bias[i], eb.biases[i]):
                    dst.copy_(torch.from_numpy(W[f"latent_codec.biases.{i}"]))
                if i < 4:
                    for dst in (m._factor[i], eb.factors[i]):
                        dst.copy_(torch.from_numpy(W[f"latent_codec.factors.{i}"]))
        v = torch.from_numpy(gi.factorized_inputs(seed)).requires_grad_(True)
        x3 = v.t().reshape(gi.H, 1, -1)
        back = lambda t: t.detach().numpy().reshape(gi.H, -1).T
        assert np.allclose(back(m._logits_cumulative(x3 - 0.5, False)), g[f"fz{seed}_lower"], rtol=1e-5, atol=1e-5)
        lik = eb._likelihood(x3)
        assert np.abs(back(lik) - g[f"fz{seed}_lik"]).max() <= 2e-7
        (lik * torch.from_numpy(g[f"fz{seed}_gw"].T.reshape(gi.H, 1, -1).copy())).sum().backward()
        assert np.allclose(v.grad.numpy(), g[f"fz{seed}_gv"], rtol=1e-3, atol=1e-7)
        for i in range(5):
            assert np.allclose(eb.matrices[i].grad.numpy(), g[f"fz{seed}_g_matrices.{i}"], rtol=1e-3, atol=1e-5)
            assert np.allclose(eb.biases[i].grad.numpy(), g[f"fz{seed}_g_biases.{i}"], rtol=1e-3, atol=1e-5)
        bits = m.forward(v.detach())                           # [N,C] semantics, bound 1e-6
        assert np.abs(bits.detach().numpy() - g[f"fz{seed}_bits"]).max() <= 1e-4
