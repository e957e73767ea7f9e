"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline): CPU restatement of the
anchor-initialisation kNN of scene/gaussian_model.py:389,407 (`simple_knn._C.distCUDA2`) and of
`voxelize_sample` (:377-380).

PARITY UNPINNED for distCUDA2: the simple_knn wheel has no source lines in /root/reference and is not installed,
so this restates its PUBLISHED behaviour — for each point the mean of the squared distances to its 3 nearest
other points, fp32 — and the HIP kernel is checked against this restatement, not against the wheel.
`voxelize_sample` IS pinned: it is three numpy calls in the mount and is restated verbatim in meaning."""
import numpy as np


def _d2(p, q):
    d = p.astype(np.float32) - q.astype(np.float32)
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def mean_dist2_brute(points: np.ndarray) -> np.ndarray:
    """O(N^2), exact, fp32 arithmetic in the kernel's operation order; small N only."""
    p = np.ascontiguousarray(points, dtype=np.float32)
    n = p.shape[0]
    out = np.empty(n, dtype=np.float32)
    big = np.float32(np.finfo(np.float32).max)
    for i in range(n):
        d = _d2(p[i][None, :], p)
        d[i] = big
        b = np.sort(d)[:3] if n >= 3 else np.concatenate([np.sort(d), np.full(3 - n, big, np.float32)])
        with np.errstate(over="ignore"):
            out[i] = ((b[0] + b[1]) + b[2]) / np.float32(3)
    return out


def mean_dist2_tree(points: np.ndarray, k_cand: int = 12) -> np.ndarray:
    """Large N: candidates from a float64 k-d tree, then the 3 smallest fp32 distances among them (excluding the
    point's own index).  k_cand > 4 absorbs ranking differences between fp64 and fp32 for near-equal distances."""
    from scipy.spatial import cKDTree
    p = np.ascontiguousarray(points, dtype=np.float32)
    n = p.shape[0]
    k = min(k_cand, n)
    _, idx = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=k)
    idx = idx.reshape(n, k)
    d = _d2(p[:, None, :], p[idx])
    d[idx == np.arange(n)[:, None]] = np.finfo(np.float32).max
    b = np.sort(d, axis=1)[:, :3]
    return ((b[:, 0] + b[:, 1]) + b[:, 2]) / np.float32(3)


def voxelize_sample(data: np.ndarray, voxel_size: float) -> np.ndarray:
    """scene/gaussian_model.py:377-380 without the in-place shuffle (np.unique sorts, so the shuffle is irrelevant)."""
    return np.unique(np.round(data / voxel_size), axis=0) * voxel_size
