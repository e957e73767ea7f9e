"""Golden vectors for the densification statistics / anchor growing from the
reference's own scene/gaussian_model.py (training_statis :696-713, anchor_growing :762-855), run on CPU in the
authoring container with the harness of tools/make_goldens.py.

Stubs that influence numbers: torch_scatter.scatter_max -> torch scatter_reduce(amax, include_self=False) (the
wheel is absent; same definition); torch.rand_like inside anchor_growing -> the seeded arrays stored in the fixture
(CPU and HIP generators differ, the draws are inputs); cat_tensors_to_optimizer -> plain concatenation (no optimizer
in this harness).  Writes tests/golden/densify.npz."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg

mg.install_stubs()
import torch_scatter


def scatter_max(src, index, dim=0):
    n = int(index.max()) + 1 if index.numel() else 0
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    out.scatter_reduce_(0, index, src, reduce="amax", include_self=False)
    return out, None


torch_scatter.scatter_max = scatter_max
sys.path.insert(0, mg.REF)
mg.patch_cuda()
out = {}
with mg.CudaToCpu():
    import scene.gaussian_model as gm
    gm.scatter_max = scatter_max
    N, K = 3000, 10
    pc = mg.build_reference_model(N, 5)
    rng = np.random.default_rng(11)
    # ---------------- training_statis ----------------
    vis = torch.from_numpy(rng.random(N) < 0.7)
    n_vis = int(vis.sum())
    opacity = torch.from_numpy(rng.normal(0.1, 0.5, size=(n_vis * K, 1)).astype(np.float32))
    sel = opacity.view(-1) > 0                                        # offset_selection_mask (neural_opacity > 0)
    P = int(sel.sum())
    update_filter = torch.from_numpy(rng.random(P) < 0.8)             # radii > 0
    grad = torch.from_numpy(rng.normal(0, 1e-3, size=(P, 3)).astype(np.float32))
    vsp = torch.zeros(P, 3, requires_grad=True)
    vsp.grad = grad.clone()
    pc.opacity_accum = torch.from_numpy(rng.random((N, 1)).astype(np.float32))
    pc.anchor_demon = torch.from_numpy(rng.integers(0, 5, (N, 1)).astype(np.float32))
    pc.offset_gradient_accum = torch.from_numpy((rng.random((N * K, 1)) * 1e-3).astype(np.float32))
    pc.offset_denom = torch.from_numpy(rng.integers(0, 5, (N * K, 1)).astype(np.float32))
    out.update(ts_vis=mg.npy(vis), ts_opacity=mg.npy(opacity), ts_sel=mg.npy(sel), ts_update_filter=mg.npy(update_filter),
               ts_grad=mg.npy(grad), ts_opacity_accum0=mg.npy(pc.opacity_accum).copy(), ts_anchor_demon0=mg.npy(pc.anchor_demon).copy(),
               ts_grad_accum0=mg.npy(pc.offset_gradient_accum).copy(), ts_denom0=mg.npy(pc.offset_denom).copy())
    pc.training_statis(vsp, opacity, update_filter, sel, vis)
    out.update(ts_opacity_accum1=mg.npy(pc.opacity_accum), ts_anchor_demon1=mg.npy(pc.anchor_demon),
               ts_grad_accum1=mg.npy(pc.offset_gradient_accum), ts_denom1=mg.npy(pc.offset_denom))
    # ---------------- anchor_growing ----------------
    grads = torch.from_numpy((rng.random(N * K) * 6e-4).astype(np.float32))
    offset_mask = torch.from_numpy(rng.random(N * K) < 0.6)
    draws, captured = [], []
    real_rand_like = torch.rand_like

    def fake_rand_like(t, *a, **k):
        r = torch.from_numpy(rng.random(tuple(t.shape)).astype(np.float32))
        draws.append(r.clone())
        return r

    def fake_cat(d):
        captured.append({k: v.clone() for k, v in d.items()})
        captured[-1]["depth"] = torch.tensor(len(draws) - 1)
        res = {}
        for name, attr in (("anchor", "_anchor"), ("scaling", "_scaling"), ("rotation", "_rotation"), ("anchor_feat", "_anchor_feat"),
                           ("hyper_latent", "_hyper_latent"), ("offset", "_offset"), ("mask", "_mask"), ("opacity", "_opacity")):
            res[name] = torch.nn.Parameter(torch.cat([getattr(pc, attr).data, d[name].float()], dim=0))
        return res

    pc.cat_tensors_to_optimizer = fake_cat
    # offsets that leave their anchor's voxel, so that some candidates survive the de-duplication at every size
    with torch.no_grad():
        pc._scaling[:, :3] = torch.log(torch.from_numpy(rng.uniform(0.05, 0.4, size=(N, 3)).astype(np.float32)))
    out.update(ag_anchor=mg.npy(pc._anchor).copy(), ag_offset=mg.npy(pc._offset).copy(), ag_scaling=mg.npy(pc._scaling).copy(),
               ag_feat=mg.npy(pc._anchor_feat).copy(), ag_hyper=mg.npy(pc._hyper_latent).copy(), ag_bound_min=mg.npy(pc.x_bound_min), ag_bound_max=mg.npy(pc.x_bound_max),
               ag_grads=mg.npy(grads), ag_offset_mask=mg.npy(offset_mask), ag_threshold=np.float32(2e-4))
    torch.rand_like = fake_rand_like
    gm.torch.rand_like = fake_rand_like
    pc.anchor_growing(grads, 2e-4, offset_mask)
    torch.rand_like = real_rand_like
    out["ag_rounds"] = np.int64(len(captured))
    for i, r in enumerate(draws):
        out[f"ag_rand{i}"] = mg.npy(r)
    out["ag_draws"] = np.int64(len(draws))
    for i, d in enumerate(captured):
        for k in ("anchor", "scaling", "anchor_feat", "hyper_latent", "offset", "mask", "opacity", "rotation", "depth"):
            out[f"ag_r{i}_{k}"] = mg.npy(d[k])
    out["ag_final_n"] = np.int64(pc._anchor.shape[0])
path = os.path.join(mg.OUT, "densify.npz")
np.savez_compressed(path, **out)
print(path, "rounds", len(captured), [int(c["anchor"].shape[0]) for c in captured], "draws", len(draws), "final", int(out["ag_final_n"]))
