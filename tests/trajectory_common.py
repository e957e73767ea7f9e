"""What tools/make_trajectory_golden.py (the REFERENCE's loop, CPU, authoring container) and tests/test_trajectory_gpu.py (the
drop-in path on the MI355X) share: the iteration schedule, the argument overrides, the cameras and target images, and every
random draw as a function of (iteration, call index) — inputs both sides regenerate, nothing of it is stored."""
from __future__ import annotations

import math

import numpy as np
import torch

N, SEED = 3000, 2
# three windows of train.py's loop: before the step-3000 switch (with the densification round at 2995), across it, across the
# step-10000 switch (update_anchor_bound at 10000, context model + rate from 10001)
ITERATIONS = list(range(2990, 3000)) + list(range(3000, 3006)) + list(range(9997, 10009))
OPT_OVERRIDES = dict(start_stat=0, update_from=2990, update_interval=5, update_until=3000, densify_grad_threshold=2e-5)
UPDATE_INIT_FACTOR = 16                   # --update_init_factor of scripts/train_mlp360.py:11 (growing grids of 16, 4, 1 voxels)
LMBDA, LMBDA_REC = 0.001, 1.0             # train.py:614-615 defaults
SPATIAL_LR_SCALE = 1.0
W, H = 80, 64
PER_ANCHOR = dict(anchor="_anchor", offset="_offset", mask="_mask", feat="_anchor_feat", hyper="_hyper_latent", scaling="_scaling")


def cameras(device="cpu"):
    from contextgs_amd.synth import orbit_cameras
    return [c.to_torch(device) for c in orbit_cameras(8, W, H)]


def gt_images(cams, device="cpu"):
    """Smooth seeded target images in [0, 1] (the reference reads viewpoint_cam.original_image, train.py:198)."""
    ys, xs = np.meshgrid(np.linspace(0, 1, H, dtype=np.float64), np.linspace(0, 1, W, dtype=np.float64), indexing="ij")
    out = []
    for i, _c in enumerate(cams):
        rng = np.random.default_rng(900 + i)
        img = np.zeros((3, H, W))
        for ch in range(3):
            for _k in range(4):
                fx, fy, ph, amp = rng.uniform(0.5, 3.0), rng.uniform(0.5, 3.0), rng.uniform(0, 2 * math.pi), rng.uniform(0.05, 0.2)
                img[ch] += amp * np.sin(2 * math.pi * (fx * xs + fy * ys) + ph)
            img[ch] += rng.uniform(0.3, 0.6)
        out.append(torch.from_numpy(np.clip(img, 0, 1).astype(np.float32)).to(device))
    return out


def seeds(it):
    """Seeds of the build's counter-based noise generator for iteration `it` (oracle.context_ref.ctx_noise / csrc/ctx_noise.h)."""
    base = 0x5EED00000000 + it * 16
    return dict(mid=base + 7, hyper=base + 3, levels=[base + 2, base + 1, base + 0])


def choose_draw(it, n):
    """The rate subset's uniform draw (scene/gaussian_model.py:1659: rand_like(anchor[:, 0]) <= 0.15)."""
    return np.random.default_rng(100_000 + it).random(n).astype(np.float32)


def grow_draw(it, k, numel):
    """The k-th rand_like of adjust_anchor at iteration `it` (anchor_growing, scene/gaussian_model.py:769)."""
    return np.random.default_rng(200_000 + it * 64 + k).random(numel).astype(np.float32)


def checksums(pc):
    """name -> (sum, sum of absolute values) in fp64 of every parameter (same names in the reference's and the drop-in model)."""
    out = {}
    for name, p in pc.named_parameters():
        d = p.detach().double()
        out[name] = (float(d.sum()), float(d.abs().sum()))
    return out
