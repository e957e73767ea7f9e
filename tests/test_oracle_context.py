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


@pytest.mark.parametrize("tag,N,seed", [("n64", 64, 1), ("n3000", 3000, 2)])
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
        bad = np.abs(a - b) > 1e-4 * step
        assert bad.mean() <= 5e-4, bad.mean()
    sums = cr.multi_scale_generating(W, anchor[mab], st["hyper"][mab], st["feat"][mab], st["offset"][mab],
                                     g["get_scaling"][mab], mask[mab], None, 0.01, ls, return_sum_bits=True,
                                     x_means=(st["feat"].mean(dtype=np.float32), g["get_scaling"].mean(dtype=np.float32),
                                              st["offset"].mean(dtype=np.float32)))
    assert sums[0] == g["msg_sum_bits"][0]
    assert np.allclose(sums[1:], g["msg_sum_bits"][1:], rtol=2e-3)


@pytest.mark.parametrize("tag,N,seed", [("n64", 64, 1), ("n3000", 3000, 2)])
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
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-4, atol=2e-6)
    # eval over the context model (:83-101)
    xyz, color, op, sc, rot, _no, _sel = cr.expand(W, g["get_anchor"][vis], g["msg_feat"][vis], g["msg_offsets"][vis],
                                                   g["msg_scaling"][vis], mask[vis], cam)
    for a, b in ((xyz, g["ev_xyz"]), (color, g["ev_color"]), (op, g["ev_opacity"]), (sc, g["ev_scaling"]), (rot, g["ev_rot"])):
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-4, atol=2e-6)
