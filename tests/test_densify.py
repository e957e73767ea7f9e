"""Densification statistics and anchor growing: the numpy oracle against the goldens
produced by the reference's own methods (CPU), and the product path (HIP statistics kernel, sort-based voxel
de-duplication) against both (GPU)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "densify.npz")
K = 10


def _z():
    return np.load(GOLD)


def test_oracle_training_statis_matches_reference():
    from oracle.densify_ref import training_statis
    z = _z()
    oa, ad, ga, dn = training_statis(K, z["ts_vis"], z["ts_opacity"], z["ts_sel"], z["ts_update_filter"], z["ts_grad"],
                                     z["ts_opacity_accum0"], z["ts_anchor_demon0"], z["ts_grad_accum0"], z["ts_denom0"])
    assert np.array_equal(ad, z["ts_anchor_demon1"]) and np.array_equal(dn, z["ts_denom1"])
    assert np.abs(oa - z["ts_opacity_accum1"]).max() < 1e-6 and np.abs(ga - z["ts_grad_accum1"]).max() < 1e-9


def _golden_rounds(z):
    return [{k: z[f"ag_r{i}_{k}"] for k in ("anchor", "scaling", "anchor_feat", "hyper_latent", "depth")}
            for i in range(int(z["ag_rounds"]))]


def _check_rounds(got, z):
    ref = _golden_rounds(z)
    assert [int(r["depth"]) for r in got] == [int(r["depth"]) for r in ref]
    for a, b in zip(got, ref):
        for k in ("anchor", "scaling", "anchor_feat", "hyper_latent"):
            x = a[k].detach().cpu().numpy() if hasattr(a[k], "detach") else np.asarray(a[k])
            assert x.shape == b[k].shape, (k, x.shape, b[k].shape)
            assert np.array_equal(x, b[k]), k            # voxel centres, log sizes and copied features: exact


def test_oracle_anchor_growing_matches_reference():
    from oracle.densify_ref import anchor_growing
    z = _z()
    rands = [z[f"ag_rand{i}"] for i in range(int(z["ag_draws"]))]
    got = anchor_growing(z["ag_anchor"], z["ag_offset"], z["ag_scaling"], z["ag_feat"], z["ag_hyper"], z["ag_bound_min"],
                         z["ag_bound_max"], z["ag_grads"], float(z["ag_threshold"]), z["ag_offset_mask"], rands, 0.01, K)
    assert len(got) == 2 and sum(r["anchor"].shape[0] for r in got) == int(z["ag_final_n"]) - z["ag_anchor"].shape[0]
    _check_rounds(got, z)


@pytest.mark.gpu
def test_hip_training_statis_matches_reference():
    import torch
    from contextgs_amd import densify
    z = _z()
    T = lambda k, dt=None: torch.tensor(z[k], device="cuda") if dt is None else torch.tensor(z[k], device="cuda", dtype=dt)

    class M:
        n_offsets = K
    pc = M()
    pc.opacity_accum, pc.anchor_demon = T("ts_opacity_accum0"), T("ts_anchor_demon0")
    pc.offset_gradient_accum, pc.offset_denom = T("ts_grad_accum0"), T("ts_denom0")
    vsp = torch.zeros(z["ts_grad"].shape, device="cuda", requires_grad=True)
    vsp.grad = T("ts_grad")
    densify.training_statis(pc, vsp, T("ts_opacity"), T("ts_update_filter"), T("ts_sel"), T("ts_vis"))
    assert torch.equal(pc.anchor_demon, T("ts_anchor_demon1")) and torch.equal(pc.offset_denom, T("ts_denom1"))
    assert float((pc.opacity_accum - T("ts_opacity_accum1")).abs().max()) < 1e-6
    assert float((pc.offset_gradient_accum - T("ts_grad_accum1")).abs().max()) < 1e-9


@pytest.mark.gpu
def test_anchor_growing_matches_reference_and_scales():
    import torch
    from contextgs_amd import densify
    z = _z()
    T = lambda k: torch.tensor(z[k], device="cuda")
    rands = [T(f"ag_rand{i}") for i in range(int(z["ag_draws"]))]
    got = densify.growing_rounds(T("ag_anchor"), T("ag_offset"), T("ag_scaling"), T("ag_feat"), T("ag_hyper"), T("ag_bound_min"),
                                 T("ag_bound_max"), T("ag_grads"), float(z["ag_threshold"]), T("ag_offset_mask"), 0.01, K,
                                 rand_fn=lambda i, like: rands[i])
    _check_rounds(got, z)
    # the de-duplication itself at scale against the reference's chunked all-pairs compare
    g = torch.Generator(device="cuda").manual_seed(0)
    grid = torch.randint(-40, 40, (200_000, 3), device="cuda", generator=g, dtype=torch.int32)
    cand = torch.unique(torch.randint(-45, 45, (30_000, 3), device="cuda", generator=g, dtype=torch.int32), dim=0)
    fast = densify.voxels_already_present(cand, grid)
    slow = torch.zeros(cand.shape[0], dtype=torch.bool, device="cuda")
    for s in range(0, grid.shape[0], 4096):
        slow |= (cand.unsqueeze(1) == grid[s:s + 4096]).all(-1).any(-1)
    assert torch.equal(fast, slow)


# ---- optimizer set-up + one full densification round against the reference's own training_setup / adjust_anchor ----
ADJ = os.path.join(os.path.dirname(__file__), "golden", "adjust_anchor.npz")
GROUPS = ("anchor", "offset", "mask", "anchor_feat", "hyper_latent", "opacity", "scaling", "rotation")


def _args(z):
    import types
    return types.SimpleNamespace(**{str(k): float(v) for k, v in zip(z["args_names"], z["args_values"])})


def test_oracle_adjust_anchor_matches_reference():
    from oracle.densify_ref import adjust_anchor
    z = np.load(ADJ)
    params = {g: z[f"pre_{g}"] for g in GROUPS}
    moments = {g: (z[f"pre_m_{g}"], z[f"pre_v_{g}"]) for g in GROUPS if bool(z[f"pre_has_state_{g}"])}
    stats = {k: z[f"pre_{k}"] for k in ("offset_denom", "offset_gradient_accum", "opacity_accum", "anchor_demon")}
    rands = [z[f"rand{i}"].reshape(-1) for i in range(int(z["draws"]))]
    p, m, s = adjust_anchor(params, moments, stats, z["bound_min"], z["bound_max"], rands, 0.01, K)
    assert p["anchor"].shape[0] == int(z["n_after"]) != int(z["n_before"])
    for g in GROUPS:
        assert np.array_equal(p[g], z[f"post_{g}"]), g
        assert (g in m) == bool(z[f"post_has_state_{g}"]), g
        if g in m:
            assert np.array_equal(m[g][0], z[f"post_m_{g}"]) and np.array_equal(m[g][1], z[f"post_v_{g}"]), g
    for k, v in s.items():
        assert np.array_equal(v, z[f"post_{k}"].astype(np.float32)), k


def test_training_setup_groups_and_learning_rate_schedule_match_reference():
    """Host logic, no kernels: the param groups, their initial learning rates and the per-iteration schedule of
    `update_learning_rate` against the reference's (OptimizationParams defaults, spatial_lr_scale 1.7)."""
    import torch
    from contextgs_amd.model import GaussianModel
    z = np.load(ADJ)
    pc = GaussianModel(device="cpu")
    n = 7
    pc.set_state(torch.zeros(n, 3), torch.zeros(n, K, 3), torch.zeros(n, K, 1), torch.zeros(n, 50), torch.zeros(n, 12),
                 torch.zeros(n, 6))
    pc.spatial_lr_scale = 1.7
    pc.training_setup(_args(z))
    assert [g["name"] for g in pc.optimizer.param_groups] == [str(s) for s in z["lr_groups"]]
    assert np.allclose([g["lr"] for g in pc.optimizer.param_groups], z["lr_initial"], rtol=1e-12, atol=0)
    assert pc.optimizer.defaults["eps"] == 1e-15
    for it, row in zip(z["lr_iterations"], z["lr_schedule"]):
        pc.update_learning_rate(int(it))
        assert np.allclose([g["lr"] for g in pc.optimizer.param_groups], row, rtol=1e-12, atol=0), int(it)
    assert pc.opacity_accum.shape == (n, 1) and pc.offset_denom.shape == (n * K, 1)


@pytest.mark.gpu
def test_adjust_anchor_matches_reference_including_the_optimizer_state():
    import torch
    from contextgs_amd.model import GaussianModel
    z = np.load(ADJ)
    T = lambda k: torch.tensor(z[k], device="cuda")
    pc = GaussianModel()
    pc.set_state(T("pre_anchor"), T("pre_offset"), T("pre_mask"), T("pre_anchor_feat"), T("pre_hyper_latent"), T("pre_scaling"))
    pc.x_bound_min, pc.x_bound_max = T("bound_min"), T("bound_max")
    pc.spatial_lr_scale = 1.7
    pc.training_setup(_args(z))
    for g in pc.optimizer.param_groups:                      # the moments the reference's two Adam steps left behind
        if g["name"] in GROUPS and bool(z[f"pre_has_state_{g['name']}"]):
            pc.optimizer.state[g["params"][0]] = {"step": torch.tensor(float(z[f"pre_step_{g['name']}"])),
                                                  "exp_avg": T(f"pre_m_{g['name']}"), "exp_avg_sq": T(f"pre_v_{g['name']}")}
    pc.offset_denom, pc.offset_gradient_accum = T("pre_offset_denom"), T("pre_offset_gradient_accum")
    pc.opacity_accum, pc.anchor_demon = T("pre_opacity_accum"), T("pre_anchor_demon")
    rands = [T(f"rand{i}").reshape(-1) for i in range(int(z["draws"]))]
    pc.adjust_anchor(check_interval=100, success_threshold=0.8, grad_threshold=2e-4, min_opacity=0.005,
                     rand_fn=lambda i, like: rands[i])
    assert pc._anchor.shape[0] == int(z["n_after"])
    attr = {"anchor": "_anchor", "offset": "_offset", "mask": "_mask", "anchor_feat": "_anchor_feat",
            "hyper_latent": "_hyper_latent", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}
    for g in pc.optimizer.param_groups:
        name = g["name"]
        if name not in GROUPS:
            continue
        p = g["params"][0]
        assert p is getattr(pc, attr[name]) and p.requires_grad == bool(z[f"post_requires_grad_{name}"]), name
        assert torch.equal(p.detach(), T(f"post_{name}")), name
        st = pc.optimizer.state.get(p, None)
        assert bool(st) == bool(z[f"post_has_state_{name}"]), name
        if st:
            assert torch.equal(st["exp_avg"], T(f"post_m_{name}")) and torch.equal(st["exp_avg_sq"], T(f"post_v_{name}")), name
            assert float(st["step"]) == float(z[f"post_step_{name}"])
    for k in ("offset_denom", "offset_gradient_accum", "opacity_accum", "anchor_demon", "max_radii2D"):
        assert torch.equal(getattr(pc, k), T(f"post_{k}").float()), k
    # the model keeps training after the surgery: one optimizer step on the new parameters
    (pc._anchor_feat.sum() + pc._offset.sum()).backward()
    pc.optimizer.step()
