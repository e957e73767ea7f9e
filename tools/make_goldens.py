#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's Python hot path on CPU.

Runs only in the authoring container (needs /root/reference, read-only).  The
reference's absent native wheels are stubbed in sys.modules, 'cuda' devices are
redirected to CPU, then the reference's own functions are called on the seeded
inputs of tests/golden_inputs.py.  Only OUTPUTS are written; the reference source
never leaves this container.

Stubs that influence numbers (declared here and in the fixtures' `_meta`):
  * compressai.entropy_models.EntropyBottleneck -> contextgs_amd.entropy_bottleneck
    (compressai is not in the mount; weights come from golden_inputs.mlp_weights).
  * torchac / diff_gaussian_rasterization / simple_knn / torch_scatter / plyfile are
    never called by the functions pinned here.

Usage: python tools/make_goldens.py
"""
from __future__ import annotations

import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------- harness -----
def install_stubs():
    from contextgs_amd import entropy_bottleneck as _eb
    from contextgs_amd.entropy_bottleneck import EntropyBottleneck
    _eb.ALLOW_HOST_FORWARD = True       # the reference runs on CPU here: the host statement of the density, on purpose

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Absent:
        def __init__(self, *a, **k):
            raise RuntimeError("stub: absent native dependency called")

    mod("torchac", encode_float_cdf=_Absent, decode_float_cdf=_Absent)
    c = mod("compressai")
    c.entropy_models = mod("compressai.entropy_models", EntropyBottleneck=EntropyBottleneck, GaussianConditional=_Absent)
    c.latent_codecs = mod("compressai.latent_codecs", LatentCodec=_Absent, HyperLatentCodec=_Absent)
    mod("plyfile", PlyData=_Absent, PlyElement=_Absent)
    s = mod("simple_knn")
    s._C = mod("simple_knn._C", distCUDA2=_Absent)
    mod("torch_scatter", scatter_max=_Absent)
    mod("diff_gaussian_rasterization", GaussianRasterizationSettings=_Absent, GaussianRasterizer=_Absent)
    mod("colorama", Fore=types.SimpleNamespace(YELLOW=""), Style=types.SimpleNamespace(RESET_ALL=""), init=lambda *a, **k: None)


class CudaToCpu(torch.overrides.TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        d = kwargs.get("device")
        if d is not None and "cuda" in str(d):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


def patch_cuda():
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda *a, **k: None


def build_reference_model(N, seed, positive_scales=False):
    import golden_inputs as gi
    from scene.gaussian_model import GaussianModel
    pc = GaussianModel(feat_dim=gi.D, n_offsets=gi.K, voxel_size=0.01, level_num=gi.LEVELS, hyper_divisor=4,
                       target_ratio=0.2)
    w = gi.mlp_weights(seed, positive_scales)
    sd = pc.state_dict()
    for k, v in w.items():
        assert k in sd, k
        sd[k] = torch.from_numpy(v)
    pc.load_state_dict(sd, strict=False)
    st = gi.anchor_state(N, seed)
    P = lambda a, g=True: torch.nn.Parameter(torch.from_numpy(a.copy()), requires_grad=g)
    pc._anchor, pc._offset, pc._mask = P(st["anchor"]), P(st["offset"]), P(st["mask"])
    pc._anchor_feat, pc._hyper_latent, pc._scaling = P(st["feat"]), P(st["hyper"]), P(st["scaling"])
    rot = np.zeros((N, 4), np.float32)
    rot[:, 0] = 1
    pc._rotation, pc._opacity = P(rot, False), P(np.zeros((N, 1), np.float32), False)
    pc.update_anchor_bound()
    return pc


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


# ---------------------------------------------------------------- goldens -----
def golden_elementwise():
    import golden_inputs as gi
    from utils.encodings import Quantize_anchor, STE_binary, STE_multistep
    from utils.entropy_models import Entropy_bernoulli, Entropy_gaussian
    out = {}
    st = gi.anchor_state(1000, 3)
    a = torch.from_numpy(st["anchor"])
    lo = a.min(0, keepdim=True)[0] * 1.2
    hi = a.max(0, keepdim=True)[0] * 1.2
    aq, q = Quantize_anchor.apply(a, lo, hi)
    out.update(qa_min=npy(lo), qa_max=npy(hi), qa_anchor_q=npy(aq), qa_quantized=npy(q))
    aq1, q1 = Quantize_anchor.apply(torch.tensor([[.1, .2, .3]]), torch.tensor([[-1., -1., -1.]]), torch.tensor([[1., 1., 1.]]))
    out.update(qa_small_q=npy(q1), qa_small_aq=npy(aq1))

    x, mean, scale, Q = (torch.from_numpy(v) for v in gi.elementwise_inputs(257, 1))
    out["ste_rowQ"] = npy(STE_multistep.apply(x, Q))
    out["ste_elemQ"] = npy(STE_multistep.apply(x, Q.expand_as(x).contiguous()))
    off = x[:, :30].reshape(-1, 10, 3).contiguous()
    out["ste_offsets"] = npy(STE_multistep.apply(off, Q.unsqueeze(1)))
    out["ste_binary"] = npy(STE_binary.apply(x / 3))

    xg, mg, sg, Qg = (t.clone().requires_grad_(True) for t in (x, mean, scale, Q))
    x_mean = torch.tensor(0.25)
    bits = Entropy_gaussian(Q=1).forward(xg, mg, sg, Qg, x_mean)
    gw = torch.from_numpy(np.random.default_rng(9).normal(size=tuple(bits.shape)).astype(np.float32))
    (bits * gw).sum().backward()
    out.update(eg_bits=npy(bits), eg_gw=npy(gw), eg_gx=npy(xg.grad), eg_gmean=npy(mg.grad), eg_gscale=npy(sg.grad),
               eg_gQ=npy(Qg.grad))
    out["eg_bits_defaultmean"] = npy(Entropy_gaussian(Q=1).forward(x, mean, scale, Q))
    out["eg_bits_scalarQ"] = npy(Entropy_gaussian(Q=0.5).forward(x, mean, scale))
    out["eb_bits"] = npy(Entropy_bernoulli().forward(torch.tensor([1., -1.]), torch.tensor([.7, .7])))
    np.savez_compressed(os.path.join(OUT, "elementwise.npz"), **out)


def golden_model(N, seed, tag, stride=1):
    import golden_inputs as gi
    import gaussian_renderer as gr
    from scene import gaussian_model as gm
    from utils.multi_level import torch_unique_with_indices
    pc = build_reference_model(N, seed)
    out = {"_meta": np.array(f"N={N} seed={seed}; EntropyBottleneck stub = contextgs_amd.entropy_bottleneck")}
    if stride > 1:      # large case: float tensors with one row per anchor / Gaussian keep every stride-th row + fp64 column sums
        out["stride"] = np.int64(stride)
    P_ = lambda name, arr: pack(out, name, arr, stride)
    with torch.no_grad():
        out.update(get_mask=npy(pc.get_mask), get_mask_anchor=npy(pc.get_mask_anchor), get_scaling=npy(pc.get_scaling),
                   get_anchor=npy(pc.get_anchor), x_bound_min=npy(pc.x_bound_min), x_bound_max=npy(pc.x_bound_max))
        anchor = pc.get_anchor
        mab = pc.get_mask_anchor.to(torch.bool)
        pc.level_scale = gm.find_divide_scale(pc, anchor[mab], pc.target_ratio, pc.level_num)
        out["level_scale"] = np.asarray(pc.level_scale, dtype=np.float64)
        key = torch.round(anchor / pc.voxel_size / pc.level_scale[0])
        u, inv, idx, cnt = torch_unique_with_indices(key, dim=0)
        out.update(uniq_rows=npy(u), uniq_inverse=npy(inv), uniq_indices=npy(idx), uniq_counts=npy(cnt))
        for variant, m in (("train", mab), ("enc", None)):
            src = anchor if m is not None else anchor[mab]
            hl, il, ml, last = gm.divide_levels(pc, src, m)
            for i in range(pc.level_num - 1):
                out[f"div_{variant}_inverse{i}"] = npy(il[i])
                out[f"div_{variant}_mapping{i}"] = npy(ml[i])
            out[f"div_{variant}_last"] = npy(last)

        pc.eval()
        f, s, o = gm.multi_scale_generating(pc, anchor, pc._hyper_latent, pc._anchor_feat, pc._offset, pc.get_scaling,
                                            pc.get_mask, mab, predict_bpp=False, training=False)
        P_("msg_feat", f), P_("msg_scaling", s), P_("msg_offsets", o)
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as td:      # return_sum_bits writes data_for_vis.pt into cwd
            os.chdir(td)
            try:
                sums = gm.multi_scale_generating(pc, anchor[mab], pc._hyper_latent[mab], pc._anchor_feat[mab],
                                                 pc._offset[mab], pc.get_scaling[mab], binary_grid_masks=pc.get_mask[mab],
                                                 predict_bpp=True, return_sum_bits=True)
            finally:
                os.chdir(cwd)
        out["msg_sum_bits"] = np.asarray(sums, dtype=np.float64)

    # expansion: eval over the context model, and training step 1000 (deterministic phase) with grads
    cam = types.SimpleNamespace(camera_center=torch.from_numpy(gi.camera_center(seed)))
    vis = torch.from_numpy(np.random.default_rng(seed + 3).random(N) < 0.8)
    out["visible_mask"] = npy(vis)
    with torch.no_grad():
        pc.eval()
        xyz, color, opacity, scaling, rot, _ = gr.generate_neural_gaussians(cam, pc, vis, is_training=False)
        for k_, t_ in (("ev_xyz", xyz), ("ev_color", color), ("ev_opacity", opacity), ("ev_scaling", scaling), ("ev_rot", rot)):
            P_(k_, t_)
        out["ev_count"] = np.int64(xyz.shape[0])
    pc.train()
    res = gr.generate_neural_gaussians(cam, pc, vis, is_training=True, step=1000)
    xyz, color, opacity, scaling, rot, neural_opacity, mask = res[:7]
    rng = np.random.default_rng(seed + 11)
    ws = [torch.from_numpy(rng.normal(size=tuple(t.shape)).astype(np.float32)) for t in (xyz, color, opacity, scaling, rot)]
    loss = sum((t * w).sum() for t, w in zip((xyz, color, opacity, scaling, rot), ws))
    loss.backward()
    for k_, t_ in (("tr_xyz", xyz), ("tr_color", color), ("tr_opacity", opacity), ("tr_scaling", scaling), ("tr_rot", rot),
                   ("tr_neural_opacity", neural_opacity), ("g_anchor", pc._anchor.grad), ("g_offset", pc._offset.grad),
                   ("g_mask", pc._mask.grad), ("g_feat", pc._anchor_feat.grad), ("g_scaling", pc._scaling.grad)):
        P_(k_, t_)
    out.update(tr_mask=npy(mask), tr_loss=np.float64(loss.item()),
               g_op_w2=npy(pc.mlp_opacity[2].weight.grad), g_cov_w0=npy(pc.mlp_cov[0].weight.grad),
               g_color_b2=npy(pc.mlp_color[2].bias.grad))
    np.savez_compressed(os.path.join(OUT, f"model_{tag}.npz"), **out)


TRAIN_SEEDS = dict(hyper_seed=0x5EED0003, seeds=[0x5EED1002, 0x5EED2001, 0x5EED3000])


def pack(out, name, arr, stride):
    """Store arr (rows strided by `stride` when > 1) plus fp64 column sums / abs-sums of ALL rows, so that a large
    per-anchor tensor is pinned without shipping every row."""
    a = npy(arr)
    out[name] = a[::stride] if stride > 1 else a
    if stride > 1:
        flat = a.reshape(a.shape[0], -1).astype(np.float64)
        out[name + "__colsum"] = flat.sum(0)
        out[name + "__abssum"] = np.abs(flat).sum(0)


def golden_training(N, seed, tag, stride, double=False):
    """double=True: the same run with every parameter, activation and accumulation in fp64 (same fp32 noise and inputs):
    the yardstick for WHICH fp32 implementation — the reference's or the HIP path — is closer on the gradient entries
    where they disagree; only the loss, the rate terms and the gradients are stored (tests/golden/train64_*.npz).

    The TRAINING variant of the context / rate path (scene/gaussian_model.py:1594-1707 with training=True,
    predict_bpp=True) as gaussian_renderer/__init__.py:63-81 calls it at step > 10000, then the expansion, with
    FIXED noise: every torch uniform_ / rand_like the reference draws is replaced by the arrays of
    oracle.context_ref.ctx_noise(seed, tensor, .) (the build's counter-based generator), so the HIP path can be
    driven with bit-identical noise.  Outputs, rate terms and the gradient of every parameter are recorded."""
    import golden_inputs as gi
    import gaussian_renderer as gr
    from scene import gaussian_model as gm
    from oracle.context_ref import ctx_noise
    pc = build_reference_model(N, seed, positive_scales=True)      # trained-like sigmas: see golden_inputs.mlp_weights
    if double:
        torch.set_default_dtype(torch.float64)
        pc.double()
        for a in ("_anchor", "_offset", "_mask", "_anchor_feat", "_hyper_latent", "_scaling", "_rotation", "_opacity"):
            old = getattr(pc, a)
            setattr(pc, a, torch.nn.Parameter(old.detach().double(), requires_grad=old.requires_grad))
        pc.x_bound_min, pc.x_bound_max = pc.x_bound_min.double(), pc.x_bound_max.double()
    pc.train()
    out = {"_meta": np.array(f"training variant, step=20000, N={N} seed={seed} stride={stride}; noise = "
                             f"oracle.context_ref.ctx_noise; EntropyBottleneck stub = contextgs_amd.entropy_bottleneck"),
           "hyper_seed": np.uint64(TRAIN_SEEDS["hyper_seed"]), "level_seeds": np.array(TRAIN_SEEDS["seeds"], np.uint64),
           "stride": np.int64(stride)}
    with torch.no_grad():
        anchor = pc.get_anchor
        mab = pc.get_mask_anchor.to(torch.bool)
        pc.level_scale = gm.find_divide_scale(pc, anchor[mab], pc.target_ratio, pc.level_num)
    out["level_scale"] = np.asarray(pc.level_scale, dtype=np.float64)
    choose_rand = np.random.default_rng(seed + 21).random(N).astype(np.float32)
    out["choose_mask"] = choose_rand <= np.float32(0.15)

    calls = {"uniform": 0, "rand": 0}
    real_uniform, real_rand_like = torch.Tensor.uniform_, torch.rand_like

    def fake_uniform_(self, a=0.0, b=1.0, **kw):
        assert (a, b) == (-0.5, 0.5), (a, b)
        k = calls["uniform"]
        calls["uniform"] += 1
        if k == 0:          # EntropyBottleneck stub: [C,1,N] view of the [N,C] hyper noise
            C = self.shape[0]
            u = ctx_noise(TRAIN_SEEDS["hyper_seed"], 3, N * C).reshape(N, C).T.reshape(C, 1, N)
        else:               # levels L-1..0, (feat, scaling, offsets) each
            lvl, t = divmod(k - 1, 3)
            u = ctx_noise(TRAIN_SEEDS["seeds"][lvl], t, self.numel()).reshape(tuple(self.shape))
        assert tuple(u.shape) == tuple(self.shape), (k, u.shape, self.shape)
        with torch.no_grad():
            self.copy_(torch.from_numpy(np.ascontiguousarray(u)))
        return self

    def fake_rand_like(t, **kw):
        calls["rand"] += 1
        assert t.shape == (N,)
        return torch.from_numpy(choose_rand.copy())

    preds = {}
    hooks = [pc.mlp_grid[i].register_forward_hook(lambda m, a, o, i=i: preds.__setitem__(i, o.detach().clone()))
             for i in range(pc.level_num)]
    cam = types.SimpleNamespace(camera_center=torch.from_numpy(gi.camera_center(seed)))
    if double:
        cam.camera_center = cam.camera_center.double()
    vis = torch.from_numpy(np.random.default_rng(seed + 3).random(N) < 0.8)
    out["visible_mask"] = npy(vis)
    torch.Tensor.uniform_, torch.rand_like = fake_uniform_, fake_rand_like
    captured = {}
    real_msg = gr.multi_scale_generating

    def spy_msg(*a, **k):
        r = real_msg(*a, **k)
        captured["msg"] = r
        return r

    gr.multi_scale_generating = spy_msg
    try:
        res = gr.generate_neural_gaussians(cam, pc, vis, is_training=True, step=20000)
    finally:
        torch.Tensor.uniform_, torch.rand_like = real_uniform, real_rand_like
        gr.multi_scale_generating = real_msg
        for h in hooks:
            h.remove()
    assert calls["uniform"] == 1 + 3 * pc.level_num and calls["rand"] == 1, calls
    (xyz, color, opacity, scaling, rot, neural_opacity, mask, bit_per_param, bpa, bit_per_feat_param, bit_per_scaling_param,
     bit_per_offsets_param, bpp_per_level) = res
    assert bpa == 16
    fq, sq, oq = captured["msg"][:3]
    pack(out, "msg_feat", fq, stride)
    pack(out, "msg_scaling", sq, stride)
    pack(out, "msg_offsets", oq.reshape(N, -1), stride)
    for i in range(pc.level_num):
        pack(out, f"pred_level{i}", preds[i], max(stride, 4))
    rng = np.random.default_rng(seed + 11)
    ws = [torch.from_numpy(rng.normal(size=tuple(t.shape)).astype(np.float32)).to(t.dtype) for t in (xyz, color, opacity, scaling, rot)]
    RW = (50.0, 30.0, 20.0, 10.0)        # weights of the four rate terms in the test loss
    loss = sum((t * w).sum() for t, w in zip((xyz, color, opacity, scaling, rot), ws))
    loss = loss + RW[0] * bit_per_param + RW[1] * bit_per_feat_param + RW[2] * bit_per_scaling_param + RW[3] * bit_per_offsets_param
    loss.backward()
    out.update(tr_mask=npy(mask), tr_loss=np.float64(loss.item()), rate_weights=np.array(RW),
               bits=np.array([bit_per_param.item(), bit_per_feat_param.item(), bit_per_scaling_param.item(),
                              bit_per_offsets_param.item()], np.float64),
               bpp_head=np.array(bpp_per_level[:2], np.float64), bpp_levels=np.array(bpp_per_level[2:], np.float64))
    for k, t in (("tr_xyz", xyz), ("tr_color", color), ("tr_opacity", opacity), ("tr_scaling", scaling), ("tr_rot", rot),
                 ("tr_neural_opacity", neural_opacity)):
        pack(out, k, t, stride)
    for k, p in (("g_anchor", pc._anchor), ("g_offset", pc._offset), ("g_mask", pc._mask), ("g_feat", pc._anchor_feat),
                 ("g_hyper", pc._hyper_latent), ("g_scaling", pc._scaling)):
        pack(out, k, p.grad.reshape(N, -1), stride)
    for name, p in pc.named_parameters():
        if name.split(".")[0] in ("mlp_opacity", "mlp_cov", "mlp_color", "mlp_grid", "latent_codec") and p.grad is not None:
            out["gw_" + name] = npy(p.grad)
    if double:
        torch.set_default_dtype(torch.float32)
        keep = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64 and k.startswith(("g_", "gw_")) and "__" not in k else v)
                for k, v in out.items() if k.startswith(("g_", "gw_", "tr_loss", "bits", "stride", "tr_mask", "_meta"))}
        keep["_meta"] = np.array(str(keep["_meta"]) + "; EVERYTHING IN FP64 (parameters .double(), default dtype float64), gradients stored rounded to fp32")
        np.savez_compressed(os.path.join(OUT, f"train64_{tag}.npz"), **keep)
        return
    np.savez_compressed(os.path.join(OUT, f"train_{tag}.npz"), **out)


def golden_entropy_api():
    """b5 / b10: the rest of utils.entropy_models + get_binary_vxl_size, and the factorised-prior DENSITY
    (Entropy_factorized._logits_cumulative + the sigmoid-difference likelihood, utils/entropy_models.py:103-135)
    with the hyper prior's weights of golden_inputs.mlp_weights — the in-mount maths that pins csrc/eb.hip."""
    import golden_inputs as gi
    from utils.encodings import get_binary_vxl_size
    from utils.entropy_models import Entropy_factorized, Entropy_gaussian_clamp, Low_bound, UniverseQuant
    out = {}
    # -- Entropy_gaussian_clamp (clamp centre = x.mean(), :8-27): values + 4 gradients
    x, mean, scale, Q = (torch.from_numpy(v).clone().requires_grad_(True) for v in gi.elementwise_inputs(193, 6))
    bits = Entropy_gaussian_clamp(Q=1).forward(x, mean, scale, Q)
    gw = torch.from_numpy(np.random.default_rng(19).normal(size=tuple(bits.shape)).astype(np.float32))
    (bits * gw).sum().backward()
    out.update(egc_bits=npy(bits), egc_gw=npy(gw), egc_gx=npy(x.grad), egc_gmean=npy(mean.grad), egc_gscale=npy(scale.grad),
               egc_gQ=npy(Q.grad))
    out["egc_bits_scalarQ"] = npy(Entropy_gaussian_clamp(Q=0.25).forward(x.detach(), mean.detach(), scale.detach()))
    # -- UniverseQuant (:159-171): a random draw per call -> statistics of the quantisation error + identity gradient
    torch.manual_seed(123)
    xu = torch.from_numpy(np.random.default_rng(20).normal(0, 3, size=(400, 250)).astype(np.float32)).requires_grad_(True)
    yu = UniverseQuant.apply(xu)
    yu.sum().backward()
    e = (yu - xu).detach().double()
    out.update(uq_err_mean=np.float64(e.mean()), uq_err_var=np.float64(e.var()), uq_err_absmax=np.float64(e.abs().max()),
               uq_grad_is_one=np.bool_(bool((xu.grad == 1).all())))
    # -- get_binary_vxl_size (utils/encodings.py:15-32)
    rng = np.random.default_rng(21)
    for k, (n, p1) in enumerate(((1000, 0.7), (30, 0.0), (30, 1.0), (77777, 0.013))):
        m = (rng.random((n, 10, 1)) < p1).astype(np.float32)
        Pg, ttl_bit, mb, ttl_num = get_binary_vxl_size(torch.from_numpy(m))
        out[f"bvs_{k}"] = np.array([Pg.item(), ttl_bit.item(), mb, ttl_num, n, p1], np.float64)
    # -- factorised prior density with the latent_codec weights (C = 12 channels, filters 3,3,3,3)
    for seed in (2, 7):
        W = gi.mlp_weights(seed)
        m = Entropy_factorized(channel=gi.H, filters=(3, 3, 3, 3))
        with torch.no_grad():
            for i in range(5):
                m._matrices[i].copy_(torch.from_numpy(W[f"latent_codec.matrices.{i}"]))
                m._bias[i].copy_(torch.from_numpy(W[f"latent_codec.biases.{i}"]))
                if i < 4:
                    m._factor[i].copy_(torch.from_numpy(W[f"latent_codec.factors.{i}"]))
        v = torch.from_numpy(gi.factorized_inputs(seed)).requires_grad_(True)             # [M, C]
        x3 = v.t().reshape(gi.H, 1, -1)                                                     # [C,1,M] as the class uses it
        lower = m._logits_cumulative(x3 - 0.5, stop_gradient=False)
        upper = m._logits_cumulative(x3 + 0.5, stop_gradient=False)
        sign = -torch.sign(torch.add(lower, upper)).detach()                               # :129-133
        lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
        bits = -torch.log2(Low_bound.apply(lik))                                          # :134-135 (bound 1e-6)
        gwl = torch.from_numpy(np.random.default_rng(seed + 30).normal(size=tuple(lik.shape)).astype(np.float32))
        (lik * gwl).sum().backward()
        back = lambda t: npy(t).reshape(gi.H, -1).T                                         # -> [M, C]
        out.update({f"fz{seed}_lower": back(lower), f"fz{seed}_upper": back(upper), f"fz{seed}_lik": back(lik),
                    f"fz{seed}_bits": back(bits), f"fz{seed}_gw": back(gwl), f"fz{seed}_gv": npy(v.grad)})
        for i in range(5):
            out[f"fz{seed}_g_matrices.{i}"] = npy(m._matrices[i].grad)
            out[f"fz{seed}_g_biases.{i}"] = npy(m._bias[i].grad)
            if i < 4:
                out[f"fz{seed}_g_factors.{i}"] = npy(m._factor[i].grad)
    np.savez_compressed(os.path.join(OUT, "entropy_api.npz"), **out)


def golden_ply_names():
    """The reference's own attribute list of the model ply (scene/gaussian_model.py:561-576, a pure-Python method) for the
    default shapes: the header our writer must produce.  (plyfile itself is absent, so no reference-written file exists.)"""
    import json
    from scene.gaussian_model import GaussianModel
    shapes = types.SimpleNamespace(_offset=torch.zeros(1, 10, 3), _mask=torch.zeros(1, 10, 1), _anchor_feat=torch.zeros(1, 50),
                                   _hyper_latent=torch.zeros(1, 12), _scaling=torch.zeros(1, 6), _rotation=torch.zeros(1, 4))
    names = GaussianModel.construct_list_of_attributes(shapes)
    with open(os.path.join(OUT, "ply_names.json"), "w") as f:
        json.dump({"n_offsets": 10, "feat_dim": 50, "hyper_dim": 12, "names": names}, f)


def main():
    assert os.path.isdir(REF), "the reference mount is required"
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    patch_cuda()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    with CudaToCpu():
        golden_elementwise()
        golden_entropy_api()
        golden_ply_names()
        golden_model(64, 1, "n64")
        golden_model(3000, 2, "n3000")
        golden_model(10000, 4, "n10000", 5)
        golden_training(3000, 2, "n3000", 1)
        golden_training(10000, 4, "n10000", 5)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
