"""Anchor-initialisation kNN: cgs_knn_mean_dist2 against oracle/knn_ref.py — BIT-EXACT
(the kernel and the oracle use the same fp32 operation order; the 3 smallest of a set do not depend on visiting order)."""
import numpy as np
import pytest
import torch

from oracle import knn_ref as ref

pytestmark = pytest.mark.gpu


def _cloud(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((n, 3), dtype=np.float32) * 4 - 2
    if kind == "clusters":          # two far, tight clusters + a few outliers: equal-count leaves must cope
        a = rng.normal(0, 0.01, (n // 2, 3)) + np.array([5, 5, 5])
        b = rng.normal(0, 0.3, (n - n // 2 - 5, 3)) - np.array([40, 0, 3])
        o = rng.uniform(-500, 500, (5, 3))
        return np.concatenate([a, b, o]).astype(np.float32)
    if kind == "dupes":             # coincident points count as neighbours at distance 0
        base = rng.random((n // 4, 3), dtype=np.float32)
        return np.concatenate([base, base, base[: n // 8], rng.random((n - 2 * (n // 4) - n // 8, 3), dtype=np.float32)])
    if kind == "plane":             # degenerate extent on one axis
        p = rng.random((n, 3), dtype=np.float32)
        p[:, 2] = 0.25
        return p
    raise ValueError(kind)


@pytest.mark.parametrize("kind,n,seed", [("uniform", 1, 0), ("uniform", 3, 1), ("uniform", 4, 2), ("uniform", 63, 3),
                                         ("uniform", 65, 4), ("uniform", 3000, 5), ("clusters", 4100, 6),
                                         ("dupes", 2048, 7), ("plane", 1500, 8)])
def test_knn_matches_brute_force_bit_exact(kind, n, seed):
    from contextgs_amd.knn import distCUDA2
    p = _cloud(kind, n, seed)
    got = distCUDA2(torch.from_numpy(p).cuda()).cpu().numpy()
    want = ref.mean_dist2_brute(p)
    assert np.array_equal(got, want), np.abs(got - want).max()


@pytest.mark.parametrize("kind,n,seed", [("uniform", 300_000, 11), ("clusters", 200_000, 12)])
def test_knn_large_matches_tree_oracle(kind, n, seed):
    from contextgs_amd.knn import distCUDA2
    p = _cloud(kind, n, seed)
    got = distCUDA2(torch.from_numpy(p).cuda()).cpu().numpy()
    assert np.array_equal(got, ref.mean_dist2_tree(p))


def test_init_from_points_matches_reference_recipe():
    """voxelisation (pinned: scene/gaussian_model.py:377-380) and the tensors create_from_pcd installs (:393-423)."""
    from contextgs_amd import knn
    rng = np.random.default_rng(3)
    pts = (rng.random((20000, 3)) * 2).astype(np.float32)
    vox = ref.voxelize_sample(pts.copy(), 0.05)
    got = knn.voxelize_sample(torch.from_numpy(pts).cuda(), 0.05).cpu().numpy()
    assert np.array_equal(got, vox)
    vs, anchor, offset, mask, feat, hyper, scaling, rot, opac = knn.init_from_points(torch.from_numpy(pts).cuda(), 0.05, 10, 50, 12)
    assert vs == 0.05 and np.array_equal(anchor.cpu().numpy(), vox.astype(np.float32))
    d2 = np.maximum(ref.mean_dist2_tree(vox.astype(np.float32)), np.float32(1e-7))
    assert np.allclose(scaling.cpu().numpy(), np.log(np.sqrt(d2))[:, None].repeat(6, 1), rtol=1e-6, atol=1e-6)
    assert offset.shape == (len(vox), 10, 3) and mask.shape == (len(vox), 10, 1) and bool((mask == 1).all())
    assert feat.shape == (len(vox), 50) and hyper.shape == (len(vox), 12) and bool((rot[:, 0] == 1).all())
    assert torch.allclose(torch.sigmoid(opac), torch.full_like(opac, 0.1))
    # voxel_size <= 0: the median kNN distance (:388-391)
    vs2 = knn.init_from_points(torch.from_numpy(pts).cuda(), 0.0, 10, 50, 12)[0]
    d = np.sort(ref.mean_dist2_tree(pts))
    assert vs2 == float(d[int(len(d) * 0.5) - 1])


def test_model_create_from_pcd_and_ply_roundtrip(tmp_path):
    """GaussianModel.create_from_pcd (:382-423) -> save_ply -> load_ply_sparse_gaussian on a second model."""
    from contextgs_amd.model import GaussianModel
    rng = np.random.default_rng(9)
    pts = (rng.random((5000, 3)) * 3).astype(np.float32)
    pc = GaussianModel(voxel_size=0.1).create_from_pcd(pts, spatial_lr_scale=2.0)
    vox = ref.voxelize_sample(pts.copy(), 0.1).astype(np.float32)
    assert np.array_equal(pc._anchor.detach().cpu().numpy(), vox) and pc.spatial_lr_scale == 2.0
    assert pc._offset.shape == (len(vox), 10, 3) and pc._scaling.shape == (len(vox), 6) and pc._opacity.shape == (len(vox), 1)
    assert not pc._rotation.requires_grad and pc._anchor.requires_grad and pc.max_radii2D.shape == (len(vox),)
    p = str(tmp_path / "point_cloud" / "iteration_1" / "point_cloud.ply")
    pc.save_ply(p)
    pc2 = GaussianModel(voxel_size=0.1).load_ply_sparse_gaussian(p)
    for name in ("_anchor", "_offset", "_mask", "_anchor_feat", "_hyper_latent", "_scaling", "_rotation", "_opacity"):
        assert torch.equal(getattr(pc, name).detach(), getattr(pc2, name).detach()), name
        assert getattr(pc2, name).requires_grad
