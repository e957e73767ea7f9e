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
    from contextgs_amd.entropy_bottleneck import EntropyBottleneck

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


def build_reference_model(N, seed):
    import golden_inputs as gi
    from scene.gaussian_model import GaussianModel
    pc = GaussianModel(feat_dim=gi.D, n_offsets=gi.K, voxel_size=0.01, level_num=gi.LEVELS, hyper_divisor=4,
                       target_ratio=0.2)
    w = gi.mlp_weights(seed)
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


def golden_model(N, seed, tag):
    import golden_inputs as gi
    import gaussian_renderer as gr
    from scene import gaussian_model as gm
    from utils.multi_level import torch_unique_with_indices
    pc = build_reference_model(N, seed)
    out = {"_meta": np.array(f"N={N} seed={seed}; EntropyBottleneck stub = contextgs_amd.entropy_bottleneck")}
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
        out.update(msg_feat=npy(f), msg_scaling=npy(s), msg_offsets=npy(o))
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
        out.update(ev_xyz=npy(xyz), ev_color=npy(color), ev_opacity=npy(opacity), ev_scaling=npy(scaling), ev_rot=npy(rot))
    pc.train()
    res = gr.generate_neural_gaussians(cam, pc, vis, is_training=True, step=1000)
    xyz, color, opacity, scaling, rot, neural_opacity, mask = res[:7]
    rng = np.random.default_rng(seed + 11)
    ws = [torch.from_numpy(rng.normal(size=tuple(t.shape)).astype(np.float32)) for t in (xyz, color, opacity, scaling, rot)]
    loss = sum((t * w).sum() for t, w in zip((xyz, color, opacity, scaling, rot), ws))
    loss.backward()
    out.update(tr_xyz=npy(xyz), tr_color=npy(color), tr_opacity=npy(opacity), tr_scaling=npy(scaling), tr_rot=npy(rot),
               tr_neural_opacity=npy(neural_opacity), tr_mask=npy(mask), tr_loss=np.float64(loss.item()),
               g_anchor=npy(pc._anchor.grad), g_offset=npy(pc._offset.grad), g_mask=npy(pc._mask.grad),
               g_feat=npy(pc._anchor_feat.grad), g_scaling=npy(pc._scaling.grad),
               g_op_w2=npy(pc.mlp_opacity[2].weight.grad), g_cov_w0=npy(pc.mlp_cov[0].weight.grad),
               g_color_b2=npy(pc.mlp_color[2].bias.grad))
    np.savez_compressed(os.path.join(OUT, f"model_{tag}.npz"), **out)


def main():
    assert os.path.isdir(REF), "the reference mount is required"
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    patch_cuda()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    with CudaToCpu():
        golden_elementwise()
        golden_model(64, 1, "n64")
        golden_model(3000, 2, "n3000")
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
