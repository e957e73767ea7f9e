"""Seeded synthetic cameras and scenes: there is no network for
datasets, so tests and bench.py run on these.  Conventions follow the
reference's Camera (scene/cameras.py:48-57) and getProjectionMatrix
(utils/graphics_utils.py:51-71): matrices are stored TRANSPOSED (row-vector
convention), znear=0.01, zfar=100.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


@dataclass
class SynthCamera:
    """Just the fields render()/prefilter_voxel() read from a viewpoint camera."""
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: object   # [4,4] row-vector convention (numpy or torch)
    full_proj_transform: object    # [4,4]
    camera_center: object          # [3]
    znear: float = 0.01
    zfar: float = 100.0

    def to_torch(self, device):
        import torch
        f = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=device)
        return SynthCamera(self.image_height, self.image_width, self.FoVx, self.FoVy,
                           f(self.world_view_transform), f(self.full_proj_transform), f(self.camera_center),
                           self.znear, self.zfar)

    def oracle_dict(self, bg=(0.0, 0.0, 0.0), scale_modifier=1.0):
        """Keyword arguments of oracle.raster_oracle.RasterOracle._cfg."""
        to_np = lambda a: a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
        return dict(H=self.image_height, W=self.image_width, tanfovx=math.tan(self.FoVx * 0.5),
                    tanfovy=math.tan(self.FoVy * 0.5), view=to_np(self.world_view_transform),
                    proj=to_np(self.full_proj_transform), bg=np.asarray(bg, dtype=np.float64),
                    scale_modifier=scale_modifier)


def projection_matrix(znear, zfar, fovx, fovy):
    """utils/graphics_utils.py:51-71 (column-vector form; caller transposes)."""
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    P = np.zeros((4, 4), dtype=np.float64)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def look_at_camera(eye, target, width, height, fovx_deg=60.0, up=(0.0, 0.0, 1.0), znear=0.01, zfar=100.0):
    eye = np.asarray(eye, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    upv = np.asarray(up, dtype=np.float64)
    right = np.cross(fwd, upv)
    if np.linalg.norm(right) < 1e-8:
        right = np.cross(fwd, np.array([0.0, 1.0, 0.0]))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    Rw2c = np.stack([right, down, fwd], axis=0)          # camera looks along +z, y down (COLMAP)
    W2C = np.eye(4)
    W2C[:3, :3] = Rw2c
    W2C[:3, 3] = -Rw2c @ eye
    fovx = math.radians(fovx_deg)
    fovy = 2 * math.atan(math.tan(fovx / 2) * height / width)
    wvt = np.float32(W2C).T.astype(np.float32)
    proj = projection_matrix(znear, zfar, fovx, fovy).T.astype(np.float32)
    full = (wvt.astype(np.float64) @ proj.astype(np.float64)).astype(np.float32)
    center = np.linalg.inv(wvt.astype(np.float64))[3, :3].astype(np.float32)
    return SynthCamera(height, width, fovx, fovy, wvt, full, center, znear, zfar)


def orbit_cameras(n, width, height, radius=3.5, fovx_deg=60.0, elevation_deg=20.0):
    """n poses on a circle of the given radius looking at the origin."""
    cams = []
    el = math.radians(elevation_deg)
    for i in range(n):
        az = 2 * math.pi * i / n
        eye = radius * np.array([math.cos(az) * math.cos(el), math.sin(az) * math.cos(el), math.sin(el)])
        cams.append(look_at_camera(eye, (0, 0, 0), width, height, fovx_deg))
    return cams


def random_gaussians(P, seed=0, extent=1.0, scale_lo=0.005, scale_hi=0.05, dtype=np.float32):
    """Free-standing Gaussians for rasterizer tests: positions in a ball, anisotropic
    scales, random rotations, colours and opacities."""
    rng = np.random.default_rng(seed)
    xyz = rng.normal(size=(P, 3))
    xyz = xyz / np.linalg.norm(xyz, axis=1, keepdims=True) * (extent * rng.random((P, 1)) ** (1 / 3))
    scales = np.exp(rng.uniform(np.log(scale_lo), np.log(scale_hi), size=(P, 3)))
    q = rng.normal(size=(P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    colors = rng.random((P, 3))
    opac = rng.uniform(0.02, 0.95, size=(P, 1))
    f = lambda a: np.ascontiguousarray(a.astype(dtype))
    return dict(means3D=f(xyz), scales=f(scales), rotations=f(q), colors=f(colors), opacities=f(opac))


def synthetic_anchors(N, voxel_size, seed=0):
    """Exactly N distinct voxel-snapped anchors: a noisy unit-sphere shell plus 20 % uniform
    in [-1,1]^3 (SURVEY §8d generator)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((0, 3), dtype=np.float64)
    factor = 1.3
    while out.shape[0] < N:
        m = int(N * factor) + 16
        n_shell = int(m * 0.8)
        d = rng.normal(size=(n_shell, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        shell = d * (1.0 + 0.05 * rng.normal(size=(n_shell, 1)))
        uni = rng.uniform(-1, 1, size=(m - n_shell, 3))
        pts = np.concatenate([shell, uni], axis=0)
        keys = np.unique(np.round(pts / voxel_size).astype(np.int64), axis=0)
        rng.shuffle(keys)
        out = keys[:N] * voxel_size
        factor *= 1.5
    return out.astype(np.float32)


def make_scene(N, seed=0, voxel_size=None, feat_dim=50, n_offsets=10, device="cuda", requires_grad=True):
    """Seeded synthetic ContextGS model with N anchors: returns a
    contextgs_amd.model.GaussianModel with random-init MLPs (there is no network for
    checkpoints) and trained-looking per-anchor state."""
    import torch
    from .model import GaussianModel

    if voxel_size is None:
        voxel_size = 0.01 if N <= 500_000 else 0.001
    rng = np.random.default_rng(seed)
    anchors = synthetic_anchors(N, voxel_size, seed)
    K, D = n_offsets, feat_dim
    base = rng.uniform(0.5, 2.0, size=(N, 6)) * voxel_size
    base[:, 3:] *= 3.0
    scaling = np.log(base)
    offset = np.clip(rng.normal(0, 0.5, size=(N, K, 3)), -2, 2)
    mask = np.where(rng.random((N, K, 1)) < 0.7, 4.0, -6.0)
    feat = np.round(rng.normal(0, 3.0, size=(N, D)))
    hyper = rng.normal(0, 2.0, size=(N, D // 4))

    torch.manual_seed(seed)
    pc = GaussianModel(feat_dim=D, n_offsets=K, voxel_size=voxel_size, device=device)
    with torch.no_grad():
        pc.mlp_opacity[2].bias.add_(0.5)          # ~60 % of the unmasked offsets survive opacity > 0
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    pc.set_state(f32(anchors), f32(offset), f32(mask), f32(feat), f32(hyper), f32(scaling),
                 requires_grad=requires_grad)
    pc.update_anchor_bound()
    return pc


class SynthPipe:
    """PipelineParams stand-in (arguments/__init__.py): the fields render() reads."""
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
