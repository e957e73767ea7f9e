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
