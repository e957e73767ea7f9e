"""Reference parity of the TRAINING-mode context / rate path (scene/gaussian_model.py:1594-1707 with
training=True, predict_bpp=True, as gaussian_renderer/__init__.py:63-81 calls it after step 10000) — the path
bench.py times every step.

tests/golden/train_*.npz hold the outputs of the REFERENCE's own Python run with FIXED noise: every uniform_ /
rand_like it draws was replaced by the build's counter-based generator (oracle.context_ref.ctx_noise, restating
csrc/ctx.hip) under recorded seeds.  Here the shipped fused HIP path is driven with the same seeds / subset, so its
noisy tensors, rate terms, per-level bpp and the gradient of EVERY parameter are compared with reference numbers,
not with a torch composition of our own.
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
CASES = [("n3000", 3000, 2), ("n10000", 10000, 4)]


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


class fixed_noise:
    """Drive the product path with the fixture's noise: the seeds of the counter-based generator in the order the step
    draws them (hyper prior first, then one per level, coarsest first — ctx_ops.next_seed) and the rate subset."""

    def __init__(self, g, N, monkeypatch):
        from contextgs_amd import context_model as cm
        from contextgs_amd import ctx_ops
        self.seeds = iter([int(g["hyper_seed"])] + [int(v) for v in g["level_seeds"]])
        choose = T(g["choose_mask"])
        monkeypatch.setattr(ctx_ops, "next_seed", lambda: next(self.seeds))
        monkeypatch.setattr(cm, "choose_mask_provider",
                            lambda anchor, mab: choose & mab if mab is not None else choose)

    def check_consumed(self):
        assert next(self.seeds, None) is None, "the hyper prior or a level did not draw its seed"


def _cmp(g, key, got, rtol, atol_of_max=0.0, outliers=0.0):
    """got (torch [N, ...]) vs fixture `key` (possibly row-strided, with fp64 column sums of all rows).
    outliers: fraction of entries allowed beyond the tolerance, each still within 2 % of the tensor's maximum."""
    a = got.detach().cpu().numpy()
    a2 = a.reshape(a.shape[0], -1)
    stride = int(g["stride"])
    ref = g[key]
    sub = a2[::stride].reshape(ref.shape) if stride > 1 else a.reshape(ref.shape)
    tol = rtol * np.abs(ref) + atol_of_max * max(1e-12, float(np.abs(ref).max()))
    bad = np.abs(sub - ref) > tol
    print(f"{key:20s} max|ref| {float(np.abs(ref).max()):10.4g}  max err {float(np.abs(sub - ref).max()):9.3g}  outside tol {int(bad.sum())}/{bad.size}")
    assert bad.mean() <= outliers, (key, int(bad.sum()), float(np.abs(sub - ref).max()), float(np.abs(ref).max()))
    assert np.abs(sub - ref).max() <= max(tol.max(), 0.02 * float(np.abs(ref).max())), key
    if stride > 1:      # all rows through their column sums
        cs, ab = a2.astype(np.float64).sum(0), g[key + "__abssum"]
        assert np.all(np.abs(cs - g[key + "__colsum"]) <= 10 * (rtol + atol_of_max) * (ab + 1e-30) + 1e-12), key


@pytest.mark.parametrize("tag,N,seed", CASES)
def test_device_noise_is_the_oracle_noise(tag, N, seed):
    """noise_quant with x = 0 and Q = q0 (1 + tanh 0) = q0 returns u q0: bit-equal to oracle.ctx_noise."""
    from contextgs_amd import ctx_ops
    from oracle.context_ref import ctx_noise
    n, D, S, O = 257, 50, 6, 30
    z = lambda w: torch.zeros(n, w, device="cuda")
    sd = int(np.load(os.path.join(GOLD, f"train_{tag}.npz"))["level_seeds"][0])
    yf, ys, yo, Q = ctx_ops.noise_quant(z(D), z(S), z(O), z(3), (1.0, 1.0, 1.0), seed=sd)
    assert np.array_equal(yf.cpu().numpy(), ctx_noise(sd, 0, n * D).reshape(n, D))
    assert np.array_equal(ys.cpu().numpy(), ctx_noise(sd, 1, n * S).reshape(n, S))
    assert np.array_equal(yo.cpu().numpy(), ctx_noise(sd, 2, n * O).reshape(n, O))
    assert float((Q - 1).abs().max()) == 0


@pytest.mark.parametrize("tag,N,seed", CASES)
def test_multi_scale_generating_training_matches_reference(tag, N, seed, monkeypatch):
    from contextgs_amd import context_model as cm
    g = np.load(os.path.join(GOLD, f"train_{tag}.npz"))
    pc = _model(N, seed)
    pc.level_scale = [float(v) for v in g["level_scale"]]
    fx = fixed_noise(g, N, monkeypatch)
    binary, mab = pc.get_mask_pair()
    res = cm.multi_scale_generating(pc, pc.get_anchor, pc._hyper_latent, pc._anchor_feat, pc._offset, pc.get_scaling,
                                    binary, mab, training=True, predict_bpp=True)
    fx.check_consumed()
    fq, sq, oq, bpp, bf, bs, bo, each = res
    # x + u Q: identical noise, Q from the fp32-MFMA level MLP -> MLP round-off (1e-6 relative) times |u| Q
    _cmp(g, "msg_feat", fq, 1e-5, 1e-6)
    _cmp(g, "msg_scaling", sq, 1e-5, 1e-6)
    _cmp(g, "msg_offsets", oq.reshape(N, -1), 1e-5, 1e-6)
    got = np.array([bpp.item(), bf.item(), bs.item(), bo.item()])
    # rate terms: means of -log2(difference of two fp32 normal CDFs) over ~0.15 N x 86 values
    assert np.allclose(got, g["bits"], rtol=1e-4), (got, g["bits"])
    each = list(each)
    assert np.allclose(each[:2], g["bpp_head"], rtol=1e-4)
    assert np.allclose(np.array(each[2:]), g["bpp_levels"], rtol=1e-4), (each[2:], g["bpp_levels"])


@pytest.mark.parametrize("tag,N,seed", CASES)
def test_training_step_outputs_and_every_gradient_match_reference(tag, N, seed, monkeypatch):
    """generate_neural_gaussians(is_training=True, step=20000) + the fixture's loss; forward values and the
    gradients of the six per-anchor tensors, all five MLPs and the hyper prior against the reference's autograd."""
    from contextgs_amd.renderer import generate_neural_gaussians
    g = np.load(os.path.join(GOLD, f"train_{tag}.npz"))
    pc = _model(N, seed)
    pc.level_scale = [float(v) for v in g["level_scale"]]
    fx = fixed_noise(g, N, monkeypatch)
    cam = types.SimpleNamespace(camera_center=T(gi.camera_center(seed)))
    res = generate_neural_gaussians(cam, pc, T(g["visible_mask"]), is_training=True, step=20000)
    fx.check_consumed()
    xyz, color, opacity, scaling, rot, neural_opacity, mask = res[:7]
    bpp, bpa, bf, bs, bo, each = res[7:]
    assert bpa == 16
    assert np.array_equal(mask.cpu().numpy(), g["tr_mask"]), "selection mask (sign of a tanh output) differs"
    for key, t in (("tr_xyz", xyz), ("tr_color", color), ("tr_opacity", opacity), ("tr_scaling", scaling), ("tr_rot", rot),
                   ("tr_neural_opacity", neural_opacity)):
        _cmp(g, key, t, 1e-4, 3e-6)
    assert np.allclose([bpp.item(), bf.item(), bs.item(), bo.item()], g["bits"], rtol=1e-4)
    each = list(each)
    assert np.allclose(each[:2], g["bpp_head"], rtol=1e-4) and np.allclose(np.array(each[2:]), g["bpp_levels"], rtol=1e-4)

    rng = np.random.default_rng(seed + 11)
    ws = [T(rng.normal(size=tuple(t.shape)).astype(np.float32)) for t in (xyz, color, opacity, scaling, rot)]
    RW = g["rate_weights"]
    loss = sum((t * w).sum() for t, w in zip((xyz, color, opacity, scaling, rot), ws))
    loss = loss + float(RW[0]) * bpp + float(RW[1]) * bf + float(RW[2]) * bs + float(RW[3]) * bo
    loss.backward()
    assert abs(loss.item() - float(g["tr_loss"])) <= 1e-4 * abs(float(g["tr_loss"])) + 1e-3
    # ---- gradients: judged against the FP64 run of the reference (tests/golden/train64_*.npz, tools/make_goldens64.py) ----
    # The reference's OWN fp32 gradients sit up to 1.5e-2 of the tensor maximum away from its fp64 gradients on the
    # entries that carry 1 / likelihood (the difference of two fp32 normal CDFs cancels there): g_scaling 1.5e-2,
    # g_offset 7e-3, g_hyper 4e-3, mlp_grid weights up to 4.5e-3 — which is why a plain "HIP vs reference fp32" comparison
    # needed an outlier allowance in round 2.  The criterion now: on every tensor the HIP path must be AS CLOSE TO THE FP64
    # TRUTH AS THE REFERENCE'S FP32 RUN IS (within a factor 2 in the max norm, floor 2e-5 of the tensor maximum for
    # tensors where both are at round-off), and no more entries may sit outside the round-2 tolerance of the fp64 value
    # than the reference itself leaves there (+2).
    g64 = np.load(os.path.join(GOLD, f"train64_{tag}.npz"))
    assert np.array_equal(g64["tr_mask"], g["tr_mask"])
    stride = int(g["stride"])

    def judge(key, got, ref32, ref64):
        big = max(float(np.abs(ref64).max()), 1e-12)
        e_hip, e_ref = np.abs(got - ref64), np.abs(ref32 - ref64)
        tol = 1e-3 * np.abs(ref64) + 1e-3 * big
        n_hip, n_ref = int((e_hip > tol).sum()), int((e_ref > tol).sum())
        print(f"{key:32s} max|ref64| {big:9.3g}  HIP-vs-fp64 {e_hip.max() / big:8.2e}  ref32-vs-fp64 {e_ref.max() / big:8.2e}"
              f"  outside 1e-3: HIP {n_hip} ref {n_ref} of {e_hip.size}")
        assert e_hip.max() <= max(2.0 * e_ref.max(), 2e-5 * big), (key, float(e_hip.max()), float(e_ref.max()), big)
        assert n_hip <= n_ref + 2, (key, n_hip, n_ref)

    for key, p in (("g_anchor", pc._anchor), ("g_offset", pc._offset), ("g_mask", pc._mask), ("g_feat", pc._anchor_feat),
                   ("g_hyper", pc._hyper_latent), ("g_scaling", pc._scaling)):
        assert p.grad is not None, key
        a = p.grad.reshape(N, -1).detach().cpu().numpy()
        sub = a[::stride].reshape(g[key].shape) if stride > 1 else a.reshape(g[key].shape)
        judge(key, sub, g[key], g64[key])
        if stride > 1:      # all rows through their fp64 column sums (reference fp32 run)
            cs, ab = a.astype(np.float64).sum(0), g[key + "__abssum"]
            assert np.all(np.abs(cs - g[key + "__colsum"]) <= 2e-2 * (ab + 1e-30) + 1e-12), key
    checked = 0
    for name, p in pc.named_parameters():
        k = "gw_" + name
        if k not in g.files:
            continue
        assert p.grad is not None, name
        a = p.grad.cpu().numpy()
        assert a.shape == g[k].shape, name
        judge(name, a, g[k], g64[k])
        checked += 1
    assert checked >= 3 * 4 + 3 * 4 + 14, checked       # anchor MLPs, level MLPs, hyper-prior matrices/biases/factors
