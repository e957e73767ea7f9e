"""Golden vectors for the optimizer set-up and one full densification round (the half round 2
left to the reference): the REFERENCE's own `training_setup` (scene/gaussian_model.py:426-525), two Adam steps on
synthetic gradients (so that the per-anchor groups carry moments), then its `adjust_anchor` (:856-910) =
anchor_growing + cat_tensors_to_optimizer + statistics bookkeeping + prune_anchor / _prune_anchor_optimizer, run on CPU
in the authoring container with the harness of tools/make_goldens.py.

Stubs that influence numbers: torch_scatter.scatter_max -> scatter_reduce(amax) (absent wheel, same definition);
torch.rand_like inside anchor_growing -> seeded arrays stored in the fixture (the draws are inputs).  The training
arguments are the reference's own OptimizationParams defaults (arguments/__init__.py:84-155).
Writes tests/golden/adjust_anchor.npz."""
import os
import sys
from argparse import ArgumentParser

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
GROUPS = ("anchor", "offset", "mask", "anchor_feat", "hyper_latent", "opacity", "scaling", "rotation")
out = {}
with mg.CudaToCpu():
    import scene.gaussian_model as gm
    from arguments import OptimizationParams
    gm.scatter_max = scatter_max
    N, K = 1200, 10
    pc = mg.build_reference_model(N, 5)
    rng = np.random.default_rng(23)
    parser = ArgumentParser()
    op = OptimizationParams(parser)
    opt_args = op.extract(parser.parse_args([]))
    out["args_names"] = np.array([k for k in sorted(vars(opt_args)) if isinstance(getattr(opt_args, k), (int, float))])
    out["args_values"] = np.array([float(getattr(opt_args, k)) for k in out["args_names"]], dtype=np.float64)
    pc.spatial_lr_scale = 1.7
    with torch.no_grad():   # offsets that leave their anchor's voxel (candidates survive), a few log-scales above the 0.05 cap
        pc._scaling[:, :3] = torch.log(torch.from_numpy(rng.uniform(0.05, 0.4, size=(N, 3)).astype(np.float32)))
        pc._scaling[:, 3:] = torch.from_numpy(rng.uniform(-3.0, 0.3, size=(N, 3)).astype(np.float32))
    pc.training_setup(opt_args)
    out["lr_groups"] = np.array([g["name"] for g in pc.optimizer.param_groups])
    out["lr_initial"] = np.array([g["lr"] for g in pc.optimizer.param_groups], dtype=np.float64)
    its = [0, 1, 500, 9999, 10000, 15000, 29999, 30000, 40000]
    sched = []
    for it in its:
        pc.update_learning_rate(it)
        sched.append([g["lr"] for g in pc.optimizer.param_groups])
    out["lr_iterations"], out["lr_schedule"] = np.array(its), np.array(sched, dtype=np.float64)
    pc.update_learning_rate(2000)
    # two Adam steps: the six differentiated per-anchor tensors get synthetic gradients, opacity / rotation / the MLPs none
    attrs = {"anchor": "_anchor", "offset": "_offset", "mask": "_mask", "anchor_feat": "_anchor_feat",
             "hyper_latent": "_hyper_latent", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}
    for step in range(2):
        for name in ("anchor", "offset", "mask", "anchor_feat", "hyper_latent", "scaling"):
            p = getattr(pc, attrs[name])
            g = torch.from_numpy(rng.normal(0, 1e-2, size=tuple(p.shape)).astype(np.float32))
            p.grad = g
        pc.optimizer.step()
        pc.optimizer.zero_grad(set_to_none=True)
    for g in pc.optimizer.param_groups:
        if g["name"] in GROUPS:
            st = pc.optimizer.state.get(g["params"][0], None)
            out[f"pre_has_state_{g['name']}"] = np.bool_(st is not None and len(st) > 0)
            out[f"pre_{g['name']}"] = mg.npy(g["params"][0]).copy()
            if st:
                out[f"pre_m_{g['name']}"], out[f"pre_v_{g['name']}"] = mg.npy(st["exp_avg"]).copy(), mg.npy(st["exp_avg_sq"]).copy()
                out[f"pre_step_{g['name']}"] = np.float64(float(st["step"]))
    # statistics of ~100 iterations: about half of the offsets seen often enough, a fifth of the anchors below the floor
    pc.offset_denom = torch.from_numpy(rng.integers(0, 90, (N * K, 1)).astype(np.float32))
    pc.offset_gradient_accum = torch.from_numpy((rng.random((N * K, 1)) * 6e-4).astype(np.float32)) * pc.offset_denom
    pc.anchor_demon = torch.from_numpy(rng.integers(40, 130, (N, 1)).astype(np.float32))
    low = torch.from_numpy(rng.random((N, 1)) < 0.25)
    pc.opacity_accum = torch.where(low, 0.001 * pc.anchor_demon, torch.from_numpy(rng.uniform(0.5, 40, (N, 1)).astype(np.float32)))
    for k in ("offset_denom", "offset_gradient_accum", "anchor_demon", "opacity_accum"):
        out[f"pre_{k}"] = mg.npy(getattr(pc, k)).copy()
    draws = []
    real_rand_like = torch.rand_like

    def fake_rand_like(t, *a, **k):
        r = torch.from_numpy(rng.random(tuple(t.shape)).astype(np.float32))
        draws.append(r.clone())
        return r

    torch.rand_like = fake_rand_like
    gm.torch.rand_like = fake_rand_like
    with torch.no_grad():
        pc.adjust_anchor(check_interval=100, success_threshold=0.8, grad_threshold=2e-4, min_opacity=0.005)
    torch.rand_like = real_rand_like
    out["draws"] = np.int64(len(draws))
    for i, r in enumerate(draws):
        out[f"rand{i}"] = mg.npy(r)
    for g in pc.optimizer.param_groups:
        if g["name"] in GROUPS:
            p = g["params"][0]
            assert p is getattr(pc, attrs[g["name"]]), g["name"]
            st = pc.optimizer.state.get(p, None)
            out[f"post_{g['name']}"] = mg.npy(p)
            out[f"post_requires_grad_{g['name']}"] = np.bool_(p.requires_grad)
            out[f"post_has_state_{g['name']}"] = np.bool_(st is not None and len(st) > 0)
            if st:
                out[f"post_m_{g['name']}"], out[f"post_v_{g['name']}"] = mg.npy(st["exp_avg"]), mg.npy(st["exp_avg_sq"])
                out[f"post_step_{g['name']}"] = np.float64(float(st["step"]))
    for k in ("offset_denom", "offset_gradient_accum", "anchor_demon", "opacity_accum", "max_radii2D"):
        out[f"post_{k}"] = mg.npy(getattr(pc, k))
    out["n_before"], out["n_after"] = np.int64(N), np.int64(pc._anchor.shape[0])
    out["bound_min"], out["bound_max"] = mg.npy(pc.x_bound_min), mg.npy(pc.x_bound_max)
path = os.path.join(mg.OUT, "adjust_anchor.npz")
np.savez_compressed(path, **out)
print(path, "draws", len(draws), "anchors", N, "->", int(out["n_after"]), "kept of the originals:",
      int((out["post_anchor"][:, None, :] == out["pre_anchor"][None, :64, :]).all(-1).any(0).sum()), "/ 64 sampled",
      os.path.getsize(path) // 1024, "KiB")
