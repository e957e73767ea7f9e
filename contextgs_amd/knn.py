"""Anchor initialisation (scene/gaussian_model.py:377-423): `distCUDA2` of the simple_knn
wheel and the voxelisation of the input cloud, on the HIP device through libcgs_hip.so (csrc/knn.hip).  There is
no CPU implementation here; oracle/knn_ref.py is the checker."""
from __future__ import annotations

import torch

from . import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """simple_knn._C.distCUDA2: mean squared distance of every point to its 3 nearest other points, float32 [N]."""
    L = _lib.lib()
    _lib.require_device(points)
    p = points.detach()
    if p.dtype != torch.float32:
        p = p.float()
    p = p.contiguous()
    assert p.dim() == 2 and p.shape[1] == 3, "points must be [N, 3]"
    n = p.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=p.device)
    if n == 0:
        return out
    ws = torch.empty(L.cgs_knn_scratch_bytes(n), dtype=torch.uint8, device=p.device)
    _lib.check(L.cgs_knn_mean_dist2(_lib.ptr(p), n, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
               "cgs_knn_mean_dist2")
    return out


def voxelize_sample(points: torch.Tensor, voxel_size: float) -> torch.Tensor:
    """GaussianModel.voxelize_sample (:377-380) on the device: unique rows of round(points / voxel) times voxel, in
    the lexicographic row order np.unique(axis=0) returns (the reference's shuffle before it does not change the
    set).  Computed in the dtype of `points`, like the numpy original."""
    q = torch.round(points / voxel_size)
    return torch.unique(q, dim=0) * voxel_size


def init_from_points(points: torch.Tensor, voxel_size: float, n_offsets: int, feat_dim: int, hyper_dim: int):
    """The tensors create_from_pcd installs (:385-423): returns (voxel_size, anchor, offset, mask, feat, hyper,
    scaling, rotation, opacity); voxel_size <= 0 selects the median kNN distance like the reference (:388-391)."""
    dev = points.device
    if voxel_size <= 0:
        d = distCUDA2(points.float())
        voxel_size = float(torch.kthvalue(d, int(d.shape[0] * 0.5)).values)
    anchor = voxelize_sample(points, voxel_size).float()
    n = anchor.shape[0]
    dist2 = torch.clamp_min(distCUDA2(anchor), 0.0000001)
    scaling = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 6)
    rot = torch.zeros(n, 4, device=dev)
    rot[:, 0] = 1
    opacity = torch.log(torch.full((n, 1), 0.1, device=dev) / (1 - 0.1))          # inverse_sigmoid(0.1)
    return (voxel_size, anchor, torch.zeros(n, n_offsets, 3, device=dev), torch.ones(n, n_offsets, 1, device=dev),
            torch.zeros(n, feat_dim, device=dev), torch.zeros(n, hyper_dim, device=dev), scaling, rot, opacity)
