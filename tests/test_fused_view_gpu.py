"""The training view with the anchor expansion fused into the rasterizer's per-Gaussian stages (csrc/expand_raster.hip,
renderer._ExpandRasterize) against the two-node path (renderer._ExpandGaussians + rasterizer._RasterizeGaussians): the same
device functions run on the same fp32 values, so everything the forward produces is bit-identical; gradients agree to the
run-to-run noise of the blend backward's float atomics (the two-node path differs from ITSELF by as much)."""
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu


def _step(fuse, step_sem, N=30000, seed=0, hw=(270, 480), loss_on_scaling=True, want_opacity_grad=False):
    from contextgs_amd import ctx_ops, renderer
    from contextgs_amd.renderer import prefilter_voxel, render
    from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
    torch.manual_seed(seed)
    ctx_ops._seed_counter = itertools.count(1)           # the same noise / rate-subset streams in every run of this test
    pc = make_scene(N, seed=seed)
    pc.train()
    cam = orbit_cameras(4, hw[1], hw[0])[1].to_torch("cuda")
    pipe, bg = SynthPipe(), torch.tensor([0.1, 0.2, 0.3], device="cuda")
    g = torch.Generator(device="cuda").manual_seed(7)
    w = torch.randn(3, hw[0], hw[1], device="cuda", generator=g)
    prev, renderer.FUSE_VIEW = renderer.FUSE_VIEW, fuse
    try:
        vis = prefilter_voxel(cam, pc, pipe, bg)
        pkg = render(cam, pc, pipe, bg, visible_mask=vis, retain_grad=True, step=step_sem)
        loss = (pkg["render"] * w).sum()
        if loss_on_scaling:
            loss = loss + 0.01 * pkg["scaling"].prod(dim=1).mean()
        if want_opacity_grad:
            loss = loss + 0.1 * pkg["neural_opacity"].clamp(min=0).sum()
        if pkg["bit_per_param"] is not None:
            loss = loss + 0.001 * pkg["bit_per_param"]
        loss.backward()
    finally:
        renderer.FUSE_VIEW = prev
    grads = {k: p.grad.detach().clone() for k, p in pc.named_parameters() if p.grad is not None}
    out = dict(image=pkg["render"].detach(), radii=pkg["radii"], scaling=pkg["scaling"].detach(), vis=pkg["visibility_filter"],
               sel=pkg["selection_mask"], no=pkg["neural_opacity"].detach(), vs_grad=pkg["viewspace_points"].grad.detach().clone(),
               bpp=None if pkg["bit_per_param"] is None else pkg["bit_per_param"].detach().clone())
    return out, grads


@pytest.mark.parametrize("step_sem,kw", [(1000, {}), (1000, dict(loss_on_scaling=False)), (1000, dict(want_opacity_grad=True)),
                                         (5000, {}), (20000, {}), (20000, dict(N=3000, hw=(64, 96)))])
def test_fused_view_equals_the_two_nodes(step_sem, kw):
    a, ga = _step(False, step_sem, **kw)
    b, gb = _step(True, step_sem, **kw)
    for k in ("image", "radii", "scaling", "vis", "sel", "no"):
        assert torch.equal(a[k], b[k]), k
    assert (a["bpp"] is None) == (b["bpp"] is None)
    if a["bpp"] is not None:                    # (the rate sums are accumulated with float atomics: equal up to their order)
        assert abs(float(a["bpp"]) - float(b["bpp"])) <= 1e-5 * abs(float(a["bpp"]))
    a2, ga2 = _step(False, step_sem, **kw)      # the two-node path against itself: the noise floor of the atomics
    assert set(ga) == set(gb) == set(ga2) and len(ga) >= 8
    vs_floor = float((a2["vs_grad"] - a["vs_grad"]).abs().max())
    assert float((b["vs_grad"] - a["vs_grad"]).abs().max()) <= max(4.0 * vs_floor, 2e-5 * float(a["vs_grad"].abs().max()))
    for k in ga:
        floor = float((ga2[k] - ga[k]).abs().max())
        diff = float((gb[k] - ga[k]).abs().max())
        assert diff <= max(4.0 * floor, 2e-5 * float(ga[k].abs().max()), 1e-12), (k, diff, floor)


def test_render_takes_the_fused_node_in_training_only():
    from contextgs_amd import renderer
    from contextgs_amd.renderer import prefilter_voxel, render
    from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
    seen = []
    orig = renderer._ExpandRasterize.apply
    renderer._ExpandRasterize.apply = staticmethod(lambda *a: (seen.append(1), orig(*a))[1])
    prev, renderer.FUSE_VIEW = renderer.FUSE_VIEW, True          # (the default; CGS_FUSE_VIEW=0 in the environment turns it off)
    try:
        pc = make_scene(3000, seed=1)
        cam = orbit_cameras(2, 96, 64)[0].to_torch("cuda")
        pipe, bg = SynthPipe(), torch.zeros(3, device="cuda")
        pc.train()
        pkg = render(cam, pc, pipe, bg, visible_mask=prefilter_voxel(cam, pc, pipe, bg), step=1000)
        assert seen == [1] and pkg["scaling"].shape[0] == pkg["radii"].shape[0]
        with torch.no_grad():                                     # no graph: the two-node path
            render(cam, pc, pipe, bg, visible_mask=prefilter_voxel(cam, pc, pipe, bg), step=1000)
        pc.eval()
        render(cam, pc, pipe, bg, visible_mask=prefilter_voxel(cam, pc, pipe, bg), step=1000)
        assert seen == [1]
    finally:
        renderer._ExpandRasterize.apply = orig
        renderer.FUSE_VIEW = prev


@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("step_sem", [1000, 20000])
@pytest.mark.parametrize("case", ["all_masked", "nothing_visible", "one_anchor"])
def test_degenerate_training_views(fuse, case, step_sem):
    """Edge cases of the training view, both node layouts: every offset masked out (P = 0 Gaussians from n > 0 visible
    anchors), a camera that sees no anchor (n = 0), a single visible anchor.  The image is the background, the backward
    runs, every parameter gets a (finite) gradient or none."""
    from contextgs_amd import renderer
    from contextgs_amd.renderer import prefilter_voxel, render
    from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
    torch.manual_seed(3)
    pc = make_scene(2000, seed=3)
    pc.train()
    cam = orbit_cameras(2, 96, 64)[0].to_torch("cuda")
    pipe, bg = SynthPipe(), torch.tensor([0.3, 0.2, 0.1], device="cuda")
    prev, renderer.FUSE_VIEW = renderer.FUSE_VIEW, fuse
    try:
        vis = prefilter_voxel(cam, pc, pipe, bg)
        if case == "all_masked":
            with torch.no_grad():
                pc._mask.fill_(-20.0)                    # sigmoid < 0.01 -> every offset masked (scene/gaussian_model.py:295-300)
        elif case == "nothing_visible":
            vis = torch.zeros_like(vis)
        else:
            keep = torch.nonzero(vis)[:1, 0]
            vis = torch.zeros_like(vis)
            vis[keep] = True
        if case == "all_masked" and step_sem > 10000:
            # no live anchor at all: the level division bisects on `kept cells / anchors` = 0 / 0, as the reference's
            # find_divide_scale does (scene/gaussian_model.py:1726-1749) — same exception, nothing to render
            with pytest.raises(ZeroDivisionError):
                render(cam, pc, pipe, bg, visible_mask=vis, retain_grad=True, step=step_sem)
            return
        pkg = render(cam, pc, pipe, bg, visible_mask=vis, retain_grad=True, step=step_sem)
        img = pkg["render"]
        assert img.shape == (3, 64, 96) and bool(torch.isfinite(img).all())
        if case != "one_anchor":
            assert pkg["radii"].numel() == 0 and pkg["scaling"].shape == (0, 3)
            assert torch.equal(img, bg.view(3, 1, 1).expand(3, 64, 96))
        loss = img.sum() + pkg["scaling"].sum()
        if pkg["bit_per_param"] is not None:
            assert bool(torch.isfinite(pkg["bit_per_param"]))
            loss = loss + 0.001 * pkg["bit_per_param"]
        loss.backward()
        for name, p in pc.named_parameters():
            assert p.grad is None or bool(torch.isfinite(p.grad).all()), name
    finally:
        renderer.FUSE_VIEW = prev
