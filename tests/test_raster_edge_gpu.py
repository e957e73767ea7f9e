"""Degenerate inputs of the drop-in rasterizer (diff_gaussian_rasterization.GaussianRasterizer's call contract,
gaussian_renderer/__init__.py:179-205): empty / single input, one-pixel and one-line images, Gaussians that cover the screen or
no pixel, opacity 0 and 1, float64 and non-contiguous inputs, everything behind the camera, a 4K frame, 20 000 Gaussians
on one pixel.  Finite image, backward runs, finite gradients."""
import math

import pytest
import torch

from contextgs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from contextgs_amd.synth import orbit_cameras

pytestmark = pytest.mark.gpu


def settings(W, H):
    cam = orbit_cameras(2, W, H)[0].to_torch("cuda")
    return cam, GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.tensor([0.1, 0.2, 0.3], device="cuda"), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=1, campos=cam.camera_center, prefiltered=False, debug=False)
def go(P, W, H, scale=0.05, dtype=torch.float32, noncontig=False, op=0.8, pos_scale=0.5):
    cam, rs = settings(W, H)
    g = torch.Generator(device="cuda").manual_seed(P + W)
    xyz = (torch.randn(P, 3, device="cuda", generator=g) * pos_scale).to(dtype).requires_grad_(True)
    col = torch.rand(P, 3, device="cuda", generator=g).to(dtype).requires_grad_(True)
    opa = torch.full((P, 1), op, device="cuda", dtype=dtype).requires_grad_(True)
    sc = torch.full((P, 3), scale, device="cuda", dtype=dtype).requires_grad_(True)
    rot = torch.nn.functional.normalize(torch.randn(P, 4, device="cuda", generator=g), dim=1).to(dtype).requires_grad_(True)
    m2d = torch.zeros(P, 3, device="cuda", requires_grad=True)
    args = dict(means3D=xyz, means2D=m2d, shs=None, colors_precomp=col, opacities=opa, scales=sc, rotations=rot, cov3D_precomp=None)
    if noncontig:
        big = torch.randn(P, 6, device="cuda", generator=g).requires_grad_(True)
        args["means3D"] = big[:, ::2] * pos_scale
    img, radii = GaussianRasterizer(rs)(**args)
    assert img.shape == (3, H, W) and bool(torch.isfinite(img).all()), "image"
    img.sum().backward()
    for t in (xyz, col, opa, sc, rot):
        assert t.grad is None or bool(torch.isfinite(t.grad).all())
    return img, radii


@pytest.mark.parametrize("name,kw", [
    ("P=0", dict(P=0, W=64, H=48)), ("P=1", dict(P=1, W=64, H=48)), ("1x1 image", dict(P=100, W=1, H=1)),
    ("one line", dict(P=100, W=257, H=1)), ("one column", dict(P=100, W=1, H=130)), ("ragged", dict(P=500, W=17, H=33)),
    ("screen-filling", dict(P=50, W=128, H=96, scale=50.0)), ("sub-pixel", dict(P=5000, W=128, H=96, scale=1e-6)),
    ("opacity 0", dict(P=500, W=64, H=48, op=0.0)), ("opacity 1", dict(P=500, W=64, H=48, op=1.0)),
    ("float64", dict(P=200, W=64, H=48, dtype=torch.float64)), ("non-contiguous", dict(P=200, W=64, H=48, noncontig=True)),
    ("behind / far away", dict(P=300, W=64, H=48, pos_scale=1e4)), ("4K", dict(P=20000, W=3840, H=2160, scale=0.02)),
    ("one pixel", dict(P=20000, W=64, H=48, pos_scale=1e-4))])
def test_rasterizer_on_degenerate_inputs(name, kw):
    img, radii = go(**kw)
    if name in ("P=0", "opacity 0"):
        bg = torch.tensor([0.1, 0.2, 0.3], device="cuda").view(3, 1, 1)
        assert torch.equal(img.detach(), bg.expand_as(img))
