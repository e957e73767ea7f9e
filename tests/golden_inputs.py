"""Seeded inputs shared by tools/make_goldens.py (which runs the REFERENCE on them in
the authoring container) and the parity tests (which run the oracle / HIP path on
them).  Everything comes from numpy's PCG64 so that both sides regenerate identical
arrays from (N, seed); only the reference's OUTPUTS are stored under tests/golden/.
"""
from __future__ import annotations

import numpy as np

D, K, H = 50, 10, 12          # feat_dim, n_offsets, hyper dim (arguments/__init__.py:50-51, hyper_divisor=4)
LEVELS = 3


def _linear(rng, fan_in, fan_out):
    b = 1.0 / np.sqrt(fan_in)
    return (rng.uniform(-b, b, size=(fan_out, fan_in)).astype(np.float32),
            rng.uniform(-b, b, size=(fan_out,)).astype(np.float32))


def mlp_weights(seed, positive_scales=False):
    """State dicts (numpy) for mlp_opacity / mlp_cov / mlp_color / mlp_grid[0..2] with the
    shapes of scene/gaussian_model.py:153-188, and the latent_codec parameters.

    positive_scales: shift the biases of the level MLPs' SCALE outputs so that the predicted sigmas are positive and of
    the order of the data's spread, as in a trained model.  With plain random weights half of the sigmas are negative,
    i.e. clamped to 1e-9, and most symbols sit on the rate model's 1e-6 likelihood floor, where the fp32 gradient is
    rounding noise on BOTH sides of a comparison (used by the training-mode fixtures, which compare gradients)."""
    rng = np.random.default_rng(seed + 1000)
    w = {}
    for name, out in (("mlp_opacity", K), ("mlp_cov", 7 * K), ("mlp_color", 3 * K)):
        w[f"{name}.0.weight"], w[f"{name}.0.bias"] = _linear(rng, D + 4, D)
        w[f"{name}.2.weight"], w[f"{name}.2.bias"] = _linear(rng, D, out)
    w["mlp_opacity.2.bias"] = w["mlp_opacity.2.bias"] + np.float32(0.3)
    out_dim = (D + 6 + 3 * K) * 2 + 3
    for i in range(LEVELS):
        in_dim = H + 3 if i == LEVELS - 1 else (D + 6 + 3) + H
        w[f"mlp_grid.{i}.0.weight"], w[f"mlp_grid.{i}.0.bias"] = _linear(rng, in_dim, 2 * D)
        w[f"mlp_grid.{i}.2.weight"], w[f"mlp_grid.{i}.2.bias"] = _linear(rng, 2 * D, out_dim)
        if positive_scales:         # output layout: [mean_f D | scale_f D | mean_s 6 | scale_s 6 | mean_o 3K | scale_o 3K | 3]
            b = w[f"mlp_grid.{i}.2.bias"]
            w[f"mlp_grid.{i}.2.weight"][D:2 * D] *= np.float32(0.5)
            b[D:2 * D] += np.float32(3.0)
            w[f"mlp_grid.{i}.2.weight"][2 * D + 6:2 * D + 12] *= np.float32(0.01)
            b[2 * D + 6:2 * D + 12] = np.float32(0.02)
            w[f"mlp_grid.{i}.2.weight"][2 * D + 12 + 3 * K:2 * D + 12 + 6 * K] *= np.float32(0.3)
            b[2 * D + 12 + 3 * K:2 * D + 12 + 6 * K] += np.float32(0.8)
            w[f"mlp_grid.{i}.2.weight"][2 * D:2 * D + 6] *= np.float32(0.02)      # scaling means near the data (~0.01-0.06)
            b[2 * D:2 * D + 6] = np.float32(0.02)
    # factorised prior (filters 3,3,3,3): perturbed around the standard init so that likelihoods vary
    f = (1, 3, 3, 3, 3, 1)
    scale = 10.0 ** (1.0 / 5)
    for i in range(5):
        init = np.log(np.expm1(1.0 / scale / f[i + 1]))
        w[f"latent_codec.matrices.{i}"] = (init + 0.1 * rng.normal(size=(H, f[i + 1], f[i]))).astype(np.float32)
        w[f"latent_codec.biases.{i}"] = rng.uniform(-0.5, 0.5, size=(H, f[i + 1], 1)).astype(np.float32)
        if i < 4:
            w[f"latent_codec.factors.{i}"] = (0.2 * rng.normal(size=(H, f[i + 1], 1))).astype(np.float32)
    w["latent_codec.quantiles"] = np.tile(np.array([-10.0, 0.0, 10.0], dtype=np.float32), (H, 1, 1))
    return w


def anchor_state(N, seed, voxel_size=0.01):
    """Per-anchor parameters (shapes of scene/gaussian_model.py:399-423)."""
    rng = np.random.default_rng(seed)
    pts = rng.normal(size=(4 * N, 3))
    pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * (0.3 + 0.7 * rng.random((4 * N, 1)))
    keys = np.unique(np.round(pts / voxel_size).astype(np.int64), axis=0)
    rng.shuffle(keys)
    assert keys.shape[0] >= N
    anchor = (keys[:N] * voxel_size).astype(np.float32)
    base = rng.uniform(0.5, 2.0, size=(N, 6)) * voxel_size
    base[:, 3:] *= 3.0
    st = {
        "anchor": anchor,
        "offset": np.clip(rng.normal(0, 0.5, size=(N, K, 3)), -2, 2).astype(np.float32),
        "mask": np.where(rng.random((N, K, 1)) < 0.7, 4.0, -6.0).astype(np.float32),
        "feat": np.round(rng.normal(0, 3.0, size=(N, D))).astype(np.float32) + rng.uniform(-0.3, 0.3, size=(N, D)).astype(np.float32),
        "hyper": rng.normal(0, 2.0, size=(N, H)).astype(np.float32),
        "scaling": np.log(base).astype(np.float32),
    }
    # a few fully-masked anchors so that get_mask_anchor is not all-true (Q4 path)
    st["mask"][:: 17] = -6.0
    return st


def camera_center(seed):
    rng = np.random.default_rng(seed + 77)
    c = rng.normal(size=3)
    return (c / np.linalg.norm(c) * 3.0).astype(np.float32)


def elementwise_inputs(n, seed):
    rng = np.random.default_rng(seed + 5)
    x = (np.round(rng.normal(0, 3, size=(n, D))) * 1.0 + rng.uniform(-0.4, 0.4, size=(n, D))).astype(np.float32)
    mean = rng.normal(0, 2, size=(n, D)).astype(np.float32)
    scale = np.exp(rng.normal(0, 1, size=(n, D))).astype(np.float32)
    scale[0, :5] = 1e-12                         # exercises the 1e-9 clamp
    Q = (1.0 * (1 + np.tanh(rng.normal(0, 0.5, size=(n, 1))))).clip(1e-9).astype(np.float32)
    x[1, :3] = 9e4                               # exercises the +-15000 Q clamp
    return x, mean, scale, Q


def factorized_inputs(seed, M=1537):
    """Values [M, H] fed to the factorised-prior density: N(0, 4^2) bulk (support +-10 and a little beyond) plus exact
    integers and half-integers around 0, where sign(lower + upper) flips."""
    rng = np.random.default_rng(seed + 40)
    v = rng.normal(0, 4.0, size=(M, H)).astype(np.float32)
    v[:21, 0] = np.arange(-10, 11, dtype=np.float32)
    v[:21, 1] = np.arange(-10, 11, dtype=np.float32) + np.float32(0.5)
    v[21, :] = 0
    return v
