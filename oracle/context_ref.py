"""numpy ORACLE for the Python half of the hot path: quantisers, Gaussian rate model,
level division, context model, anchor->Gaussian expansion.

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by contextgs_amd/.

Pinned: every function here is checked against tests/golden/*.npz, which hold the
outputs of the REFERENCE's own Python (imported on CPU in the authoring container by
tools/make_goldens.py) on the seeded inputs of tests/golden_inputs.py.

Each function cites the reference lines it restates (paths relative to the reference).
All arithmetic is fp32 in the reference's operation order.
"""
from __future__ import annotations

import numpy as np
from scipy.special import erf

f32 = np.float32


# ---- utils/encodings.py ---------------------------------------------------------------
def div_floor(a, b):
    """torch.div(a, b, rounding_mode='floor') for fp32 (utils/encodings.py:224)."""
    a, b = np.asarray(a, f32), np.asarray(b, f32)
    with np.errstate(all="ignore"):
        mod = np.fmod(a, b).astype(f32)
        div = ((a - mod) / b).astype(f32)
        fix = (mod != 0) & ((b < 0) != (mod < 0))
        div = np.where(fix, div - f32(1), div).astype(f32)
        fl = np.floor(div).astype(f32)
        fl = np.where(div - fl > f32(0.5), fl + f32(1), fl).astype(f32)
        zero = np.copysign(f32(0), (a / b).astype(f32))
        return np.where(div != 0, fl, zero).astype(f32)


def quantize_anchor(anchors, min_v, max_v, digits=16):
    """Quantize_anchor.forward, utils/encodings.py:219-227."""
    q_anchor = f32(1 / (2 ** digits - 1))
    anchors, min_v, max_v = (np.asarray(v, f32) for v in (anchors, min_v, max_v))
    interval = ((max_v - min_v) * q_anchor + f32(1e-6)).astype(f32)
    q = div_floor(anchors - min_v, interval)
    q = np.clip(q, f32(0), f32(2 ** digits - 1)).astype(f32)
    return (q * interval + min_v).astype(f32), q


def ste_multistep(x, Q):
    """STE_multistep.forward, utils/encodings.py:205-213 (use_clamp=True)."""
    x, Q = np.asarray(x, f32), np.asarray(Q, f32)
    x = np.minimum(np.maximum(x, f32(-15000) * Q), f32(15000) * Q).astype(f32)
    return (np.round((x / Q).astype(f32)) * Q).astype(f32)


def ste_binary(x):
    """STE_binary.forward, utils/encodings.py:185-192."""
    x = np.clip(np.asarray(x, f32), -1, 1)
    return np.where(x >= 0, f32(1), f32(-1)).astype(f32)


def binary_vxl_size(binary_vxl):
    """get_binary_vxl_size, utils/encodings.py:15-32 -> (Pg, ttl_bit, MB, ttl_num)."""
    m = np.asarray(binary_vxl).astype(np.float64)
    pos = m.sum()
    pg = np.clip(f32(pos / m.size), f32(1e-6), f32(1 - 1e-6))
    ttl_bit = float(f32(pos) * -np.log2(pg) + f32(m.size - pos) * -np.log2(f32(1) - pg) + 32)
    return float(pg), ttl_bit, ttl_bit / 8.0 / 1024 / 1024, m.size


# ---- utils/entropy_models.py ------------------------------------------------------------
def _normal_cdf(v, mean, scale):
    return (f32(0.5) * (f32(1) + erf(((v - mean) * (f32(1) / scale) / f32(np.sqrt(2.0))).astype(f32)))).astype(f32)


def entropy_gaussian(x, mean, scale, Q, x_mean=None):
    """Entropy_gaussian.forward, utils/entropy_models.py:34-50 (+ Low_bound :143-147)."""
    x, mean, scale, Q = (np.asarray(v, f32) for v in (x, mean, scale, Q))
    if x_mean is None:
        x_mean = x.mean(dtype=f32)
    x_mean = f32(x_mean)
    x = np.minimum(np.maximum(x, x_mean - f32(15000) * Q), x_mean + f32(15000) * Q).astype(f32)
    scale = np.maximum(scale, f32(1e-9))
    lower = _normal_cdf(x - f32(0.5) * Q, mean, scale)
    upper = _normal_cdf(x + f32(0.5) * Q, mean, scale)
    lik = np.maximum(np.abs(upper - lower), f32(1e-6))
    return (-np.log2(lik)).astype(f32)


def entropy_gaussian_grads(x, mean, scale, Q, x_mean, g_bits):
    """Analytic gradient of the reference's autograd graph for Entropy_gaussian incl.
    Low_bound.backward (utils/entropy_models.py:149-156), in float64."""
    x, mean, scale, Q, g = (np.asarray(v, np.float64) for v in (x, mean, scale, Q, g_bits))
    Qb = np.broadcast_to(Q, x.shape)
    lo, hi = x_mean - 15000 * Qb, x_mean + 15000 * Qb
    in_range = (x >= lo) & (x <= hi)
    xc = np.clip(x, lo, hi)
    s = np.maximum(scale, 1e-9)
    zu = (xc + 0.5 * Qb - mean) / s / np.sqrt(2)
    zl = (xc - 0.5 * Qb - mean) / s / np.sqrt(2)
    diff = 0.5 * (erf(zu) - erf(zl))
    lik = np.abs(diff)
    ok = lik >= 1e-6
    g_lik = np.where(ok, g * (-1 / np.log(2)) / np.where(ok, lik, 1), 0)
    g_diff = g_lik * np.sign(diff)
    g_zu = g_diff * np.exp(-zu ** 2) / np.sqrt(np.pi)
    g_zl = -g_diff * np.exp(-zl ** 2) / np.sqrt(np.pi)
    k = 1 / s / np.sqrt(2)
    g_xc = (g_zu + g_zl) * k
    return (np.where(in_range, g_xc, 0), -g_xc, np.where(scale >= 1e-9, -(g_zu * zu + g_zl * zl) / s, 0),
            (0.5 * (g_zu - g_zl) * k))


def entropy_bernoulli(x, p):
    """Entropy_bernoulli.forward, utils/entropy_models.py:56-64."""
    x, p = np.asarray(x, f32), np.clip(np.asarray(p, f32), f32(1e-6), f32(1 - 1e-6))
    return (-np.log2(p) * ((1 + x) / 2) + -np.log2(1 - p) * ((1 - x) / 2)).astype(f32)


# ---- utils/multi_level.py + scene/gaussian_model.py level functions ----------------------
def unique_with_indices(rows):
    """torch_unique_with_indices, utils/multi_level.py:3-31: lexicographic unique rows,
    inverse, SMALLEST original index per group (scatter_reduce amin), counts."""
    rows = np.asarray(rows, f32) + f32(0)       # -0.0 == 0.0
    u, first, inv, cnt = np.unique(rows, axis=0, return_index=True, return_inverse=True, return_counts=True)
    return u, inv.reshape(-1).astype(np.int64), first.astype(np.int64), cnt.astype(np.int64)


def voxel_key(anchor, voxel_size, scale):
    """torch.round(anchor / voxel_size / scale), fp32 left to right (scene/gaussian_model.py:1732,1760)."""
    return np.round(((np.asarray(anchor, f32) / f32(voxel_size)).astype(f32) / f32(scale)).astype(f32))


def find_divide_scale(anchor, x_bound_min, x_bound_max, voxel_size, target_ratio, level_num):
    """scene/gaussian_model.py:1726-1749."""
    scale_upper = f32(((np.asarray(x_bound_max, f32) - np.asarray(x_bound_min, f32)) / f32(voxel_size)).max())
    anchor_unique = np.asarray(anchor, f32)
    scales = []
    scale_lower = f32(1)
    for _ in range(level_num - 1):
        up, low = scale_upper, scale_lower
        while True:
            scale = f32((up + low) / f32(2))
            u = np.unique(voxel_key(anchor_unique, voxel_size, scale) + f32(0), axis=0)
            uniq = ((u * f32(voxel_size)).astype(f32) * scale).astype(f32)
            ratio = uniq.shape[0] / anchor_unique.shape[0]
            if abs(ratio - target_ratio) < 0.01 or abs(up - low) < 1:
                break
            if ratio < target_ratio:
                up = scale
            else:
                low = scale
        anchor_unique = uniq
        scale_lower = scale
        scales.append(float(scale))
    return scales


def divide_levels(anchor, voxel_size, level_scale, level_num, mask_anchor_bool=None):
    """scene/gaussian_model.py:1751-1765."""
    hybrid = np.asarray(anchor, f32)
    anchors, inverse_list, mapping_list = [hybrid], [], []
    for i in range(1, level_num):
        if i == 1 and mask_anchor_bool is not None:
            hybrid = (hybrid * mask_anchor_bool[:, None].astype(f32)).astype(f32)
        _u, inv, first, _c = unique_with_indices(voxel_key(hybrid, voxel_size, level_scale[i - 1]))
        hybrid = hybrid[first]
        anchors.append(hybrid)
        inverse_list.append(inv)
        mapping_list.append(first)
    return anchors, inverse_list, mapping_list, hybrid


def mapping_to_orign(mapping_list, L, mask=None):
    """scene/gaussian_model.py:1768-1787."""
    m = mapping_list[L - 1] if mask is None else mapping_list[L - 1][mask]
    for i in reversed(range(L - 1)):
        m = mapping_list[i][m]
    return m


def index_of_level_L_in_orign(mapping_list, inverse_list, idx, L):
    """scene/gaussian_model.py:1789-1792."""
    for i in range(L):
        idx = inverse_list[i][idx]
    return mapping_to_orign(mapping_list, L, mask=idx)


def extract_context_feat(anchor, feat_Q, scaling_Q, already_coded, inverse_list, mapping_list, i):
    """scene/gaussian_model.py:1711-1724."""
    content = np.concatenate([anchor, feat_Q, scaling_Q], axis=1)
    if i > 1:
        m = np.zeros(content.shape[0], bool)
        m[mapping_to_orign(mapping_list, i - 1)] = True
    else:
        m = np.ones(content.shape[0], bool)
    m &= ~already_coded
    idx = index_of_level_L_in_orign(mapping_list, inverse_list, np.nonzero(m)[0], i)
    return content[idx]


# ---- MLPs and the hyper prior --------------------------------------------------------------
def linear(x, w, b):
    return (x.astype(f32) @ w.T.astype(f32) + b.astype(f32)).astype(f32)


def mlp2(x, W, prefix):
    """nn.Sequential(Linear, ReLU, Linear) (scene/gaussian_model.py:153-188)."""
    h = np.maximum(linear(x, W[f"{prefix}.0.weight"], W[f"{prefix}.0.bias"]), 0)
    return linear(h, W[f"{prefix}.2.weight"], W[f"{prefix}.2.bias"])


def _softplus(x):
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20)))).astype(f32)


def bottleneck_logits(x, W):
    """Factorised-prior cumulative logits (the maths of utils/entropy_models.py:103-119);
    x [C,1,M]."""
    logits = x.astype(f32)
    for i in range(5):
        logits = (np.matmul(_softplus(W[f"latent_codec.matrices.{i}"]), logits) + W[f"latent_codec.biases.{i}"]).astype(f32)
        if i < 4:
            logits = (logits + np.tanh(W[f"latent_codec.factors.{i}"]) * np.tanh(logits)).astype(f32)
    return logits


def bottleneck_eval(hyper, W):
    """EntropyBottleneck.forward(x, training=False): dequantise about the medians + likelihood."""
    med = W["latent_codec.quantiles"][:, :, 1:2]
    v = hyper.T.reshape(hyper.shape[1], 1, -1).astype(f32)
    out = (np.round(v - med) + med).astype(f32)
    lower = bottleneck_logits(out - f32(0.5), W)
    upper = bottleneck_logits(out + f32(0.5), W)
    sign = -np.sign(lower + upper)
    sig = lambda t: (1 / (1 + np.exp(-t))).astype(f32)
    lik = np.maximum(np.abs(sig(sign * upper) - sig(sign * lower)), f32(1e-9))
    back = lambda t: t.reshape(hyper.shape[1], -1).T
    return back(out), back(lik)


def bottleneck_train(hyper, W, noise):
    """EntropyBottleneck.forward(x, training=True): x + U(-1/2, 1/2) noise (given, [N,C]) + likelihood."""
    v = (hyper.astype(f32) + noise.astype(f32)).astype(f32)
    lik = bottleneck_likelihood(v, W)
    return v, lik


def bottleneck_likelihood(v, W):
    """|sigmoid(s u) - sigmoid(s l)| with s = -sign(l + u), floored at 1e-9, of values v [N,C] (the density of
    utils/entropy_models.py:121-138 with Q = 1 and compressai's 1e-9 bound)."""
    x = v.T.reshape(v.shape[1], 1, -1).astype(f32)
    lower = bottleneck_logits(x - f32(0.5), W)
    upper = bottleneck_logits(x + f32(0.5), W)
    sign = -np.sign(lower + upper)
    sig = lambda t: (1 / (1 + np.exp(-t))).astype(f32)
    lik = np.maximum(np.abs(sig(sign * upper) - sig(sign * lower)), f32(1e-9))
    return lik.reshape(v.shape[1], -1).T


def _mix32(x):
    """lowbias32 integer mixer on uint32 arrays (wrap-around multiplication)."""
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint32(16)); x = x * np.uint32(0x7feb352d)
        x = x ^ (x >> np.uint32(15)); x = x * np.uint32(0x846ca68b)
        x = x ^ (x >> np.uint32(16))
    return x


def ctx_noise(seed, tensor, count):
    """u in [-0.5, 0.5) for element 0..count-1 of tensor `tensor` (0 feat, 1 scaling, 2 offsets, 3 hyper) of the
    counter-based training noise stream `seed`:  key = mix32(seed_lo ^ golden32 * (tensor + 1)) ^ seed_hi,
    u = (mix32(e_lo + key + e_hi * c) >> 8) / 2^24 - 1/2.
    NOT a reference function (the reference draws torch.uniform_, scene/gaussian_model.py:1610-1616): this restates
    the build's own generator (csrc/ctx.hip ctx_noise) so that fixtures can feed the SAME noise to the reference."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        key = _mix32(np.array([(seed & 0xFFFFFFFF) ^ ((0x9E3779B9 * (tensor + 1)) & 0xFFFFFFFF)], np.uint32))[0] ^ np.uint32(seed >> 32)
        e = np.arange(count, dtype=np.uint64)
        lo, hi = (e & np.uint64(0xFFFFFFFF)).astype(np.uint32), (e >> np.uint64(32)).astype(np.uint32)
        h = _mix32(lo + key + hi * np.uint32(0x632BE5AB))
    return ((h >> np.uint32(8)).astype(f32) * f32(1.0 / 16777216.0) - f32(0.5)).astype(f32)


# ---- scene/gaussian_model.py:1541-1707 ------------------------------------------------------------
def multi_scale_generating(W, anchor, hyper, feat, offsets, scaling, masks, mask_anchor_bool, voxel_size, level_scale,
                           level_num=3, D=50, K=10, return_sum_bits=False, x_means=None, train=None):
    """Eval variants by default.  train = dict(seeds=[one per coded level, L-1 first], hyper_seed=int,
    choose_mask=bool[N]) selects the TRAINING variant (training=True, predict_bpp=True, :1610-1616, :1686-1705)
    with the noise of ctx_noise() and returns (feat_Q, scaling_Q, offsets_Q, rates dict, per-level stats)."""
    n = anchor.shape[0]
    feat_Q, scaling_Q = np.zeros_like(feat), np.zeros_like(scaling)
    offsets_Q = np.zeros_like(offsets)
    already = np.zeros(n, bool)
    stats = {k: np.zeros((n, d), f32) for k, d in (("mf", D), ("sf", D), ("qf", 1), ("ms", 6), ("ss", 6), ("qs", 1),
                                                    ("mo", 3 * K), ("so", 3 * K), ("qo", 1))}
    if train is None:
        hyper_feat, lik_hyper = bottleneck_eval(hyper, W)
    else:
        hyper_feat, lik_hyper = bottleneck_train(hyper, W, ctx_noise(train["hyper_seed"], 3, hyper.size).reshape(hyper.shape))
    anchors_l, inverse_list, mapping_list, _last = divide_levels(anchor, voxel_size, level_scale, level_num, mask_anchor_bool)
    context = None
    coded, level_rows, preds = 0, [], []
    for i in reversed(range(level_num)):
        n_level = n if i == 0 else mapping_list[i - 1].shape[0]
        to_code = np.ones(n_level, bool)
        if i != level_num - 1:
            to_code[mapping_list[i]] = False
        orig = mapping_to_orign(mapping_list, i, to_code) if i != 0 else np.arange(n)[to_code]
        if orig.shape[0] > 0:
            lvl_anchor = anchors_l[i][to_code]
            x_in = np.concatenate([lvl_anchor if context is None else context, hyper_feat[orig]], axis=1)
            pred = mlp2(x_in, W, f"mlp_grid.{i}")
            sp = np.cumsum([D, D, 6, 6, 3 * K, 3 * K, 1, 1])
            mf, sf, ms, ss, mo, so, aqf, aqs, aqo = np.split(pred, sp, axis=1)
            Qf = np.maximum(f32(1) * (1 + np.tanh(aqf)), f32(1e-9)).astype(f32)
            Qs = np.maximum(f32(0.001) * (1 + np.tanh(aqs)), f32(1e-9)).astype(f32)
            Qo = np.maximum(f32(0.2) * (1 + np.tanh(aqo)), f32(1e-9)).astype(f32)
            if train is None:
                feat_Q[orig] = ste_multistep(feat[orig], Qf)
                scaling_Q[orig] = ste_multistep(scaling[orig], Qs)
                offsets_Q[orig] = ste_multistep(offsets[orig], Qo[:, None, :])
            else:
                sd, m = train["seeds"][coded], orig.shape[0]
                feat_Q[orig] = (feat[orig] + ctx_noise(sd, 0, m * D).reshape(m, D) * Qf).astype(f32)
                scaling_Q[orig] = (scaling[orig] + ctx_noise(sd, 1, m * 6).reshape(m, 6) * Qs).astype(f32)
                offsets_Q[orig] = (offsets[orig] + ctx_noise(sd, 2, m * 3 * K).reshape(m, K, 3) * Qo[:, None, :]).astype(f32)
            coded += 1
            level_rows.append(orig)
            preds.append(pred)
            for k, v in (("mf", mf), ("sf", sf), ("qf", Qf), ("ms", ms), ("ss", ss), ("qs", Qs), ("mo", mo), ("so", so), ("qo", Qo)):
                stats[k][orig] = v
            already[orig] = True
        if i != 0:
            context = extract_context_feat(anchor, feat_Q, scaling_Q, already, inverse_list, mapping_list, i)
    if not return_sum_bits and train is None:
        return feat_Q, scaling_Q, offsets_Q
    # :1657-1685 with chosse_random_thresh = 1 (every anchor chosen) / the given subset in training
    sel = np.ones(n, bool) if train is None else train["choose_mask"].copy()
    if mask_anchor_bool is not None:
        sel &= mask_anchor_bool
    bit_hyper = -np.log2(lik_hyper[sel])
    bf = entropy_gaussian(feat_Q[sel], stats["mf"][sel], stats["sf"][sel], stats["qf"][sel], x_means[0])
    bs = entropy_gaussian(scaling_Q[sel], stats["ms"][sel], stats["ss"][sel], stats["qs"][sel], x_means[1])
    bo = entropy_gaussian(offsets_Q[sel].reshape(-1, 3 * K), stats["mo"][sel], stats["so"][sel], stats["qo"][sel], x_means[2])
    bo = bo * np.tile(masks[sel], (1, 1, 3)).reshape(-1, 3 * K)
    if train is not None:                                   # :1687-1705
        f64 = np.float64
        rate = f64(mask_anchor_bool.sum()) / mask_anchor_bool.size if mask_anchor_bool is not None else 1.0
        S = lambda a: a.sum(dtype=f64)
        rates = dict(bit_per_param=(S(bf) + S(bs) + S(bo) + S(bit_hyper)) / (bf.size + bs.size + bo.size) * rate,
                     bit_per_feat_param=S(bf) / bf.size * rate, bit_per_scaling_param=S(bs) / bs.size * rate,
                     bit_per_offsets_param=S(bo) / bo.size * rate, bit_per_hyper_param=S(bit_hyper) / bit_hyper.size * rate)
        bpp_map = bo.sum(1, dtype=f64) + bs.sum(1, dtype=f64) + bf.sum(1, dtype=f64)
        each = [1 - (mask_anchor_bool.astype(f32).mean() if mask_anchor_bool is not None else 1.0), rates["bit_per_hyper_param"]]
        for orig in level_rows:
            lm = np.zeros(n, bool)
            lm[orig] = True
            each.append([orig.shape[0] / n, bpp_map[lm[sel]].mean() / (D + 6 + 3 * K)])
        rates["each_level_bpp"] = each
        return feat_Q, scaling_Q, offsets_Q, rates, dict(level_rows=level_rows, preds=preds, hyper_feat=hyper_feat)
    bit_masks = binary_vxl_size(masks)[1]
    return (bit_hyper.shape[0] * 3 * 16, float(bit_hyper.sum(dtype=np.float64)), float(bf.sum(dtype=np.float64)),
            float(bs.sum(dtype=np.float64)), float(bo.sum(dtype=np.float64)), bit_masks)


# ---- gaussian_renderer/__init__.py:106-145 -------------------------------------------------------
def expand(W, anchor, feat, offsets, scaling, masks, cam_center, D=50, K=10):
    """Anchor -> Gaussians for already-selected (visible) anchors. Returns the compacted
    (xyz, color, opacity, scaling, rot) plus neural_opacity and the selection mask."""
    ob_view = (anchor - cam_center).astype(f32)
    ob_dist = np.sqrt((ob_view.astype(f32) ** 2).sum(1, keepdims=True, dtype=f32)).astype(f32)
    ob_view = (ob_view / ob_dist).astype(f32)
    x = np.concatenate([feat, ob_view, ob_dist], axis=1).astype(f32)
    no = np.tanh(mlp2(x, W, "mlp_opacity")).reshape(-1, 1).astype(f32)
    no = (no * masks.reshape(-1, 1)).astype(f32)
    sel = (no > 0).reshape(-1)
    color = (1 / (1 + np.exp(-mlp2(x, W, "mlp_color")))).astype(f32).reshape(-1, 3)
    sr = mlp2(x, W, "mlp_cov").reshape(-1, 7)
    rep = lambda a: np.repeat(a, K, axis=0)
    gs, anc, off = rep(scaling)[sel], rep(anchor)[sel], offsets.reshape(-1, 3)[sel]
    sr, color = sr[sel], color[sel]
    sc = (gs[:, 3:] * (1 / (1 + np.exp(-sr[:, :3])))).astype(f32)
    q = sr[:, 3:7]
    rot = (q / np.maximum(np.sqrt((q ** 2).sum(1, keepdims=True)), f32(1e-12))).astype(f32)
    xyz = (anc + off * gs[:, :3]).astype(f32)
    return xyz, color, no[sel], sc, rot, no, sel
