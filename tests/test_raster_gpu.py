"""GPU parity: HIP rasterizer (through the C-ABI of include/cgs.h) vs the CPU oracle.

Tolerances: the image must match the fp32 oracle to RMSE <= 1e-5 (north_star: 1e-4
PSNR-equivalent); isolated pixels may differ by one alpha>=1/255 decision flipping on the
last ulp (a discontinuity of the algorithm itself), so max-abs is checked on all but a
1e-4 fraction of the pixels.  Gradients: atomics re-associate sums, tolerance 2e-4 relative
to the tensor's max magnitude.  Integer outputs (radii, sort, scan) are bit-exact.
"""
import math

import numpy as np
import pytest
import torch

from contextgs_amd.synth import look_at_camera, orbit_cameras, random_gaussians

pytestmark = pytest.mark.gpu


def _settings(cam, bg, scale_modifier=1.0, debug=False):
    from contextgs_amd.rasterizer import GaussianRasterizationSettings
    c = cam.to_torch("cuda")
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width,
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.tensor(bg, dtype=torch.float32, device="cuda"), scale_modifier=scale_modifier,
        viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform, sh_degree=1,
        campos=c.camera_center, prefiltered=False, debug=debug)


def _run_gpu(cam, g, bg, w=None, scale_modifier=1.0):
    from contextgs_amd.rasterizer import GaussianRasterizer
    t = {k: torch.tensor(v, device="cuda", requires_grad=(w is not None)) for k, v in g.items()}
    means2D = torch.zeros_like(t["means3D"], requires_grad=(w is not None))
    rast = GaussianRasterizer(_settings(cam, bg, scale_modifier, debug=True))
    color, radii = rast(means3D=t["means3D"], means2D=means2D, shs=None, colors_precomp=t["colors"],
                        opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    out = {"color": color.detach().cpu().numpy(), "radii": radii.cpu().numpy()}
    if w is not None:
        (color * torch.tensor(w, device="cuda")).sum().backward()
        out.update(dL_dmeans3D=t["means3D"].grad, dL_dmeans2D=means2D.grad, dL_dcolors=t["colors"].grad,
                   dL_dopacities=t["opacities"].grad.reshape(-1), dL_dscales=t["scales"].grad,
                   dL_drotations=t["rotations"].grad)
        for k in list(out):
            if k.startswith("dL_"):
                out[k] = out[k].cpu().numpy()
    return out


def _kw(g):
    return dict(means3D=g["means3D"], colors=g["colors"], opacities=g["opacities"], scales=g["scales"],
                rots=g["rotations"])


ALLOWANCE = []      # (what, entries inside the allowance, entries, worst): printed at the end of the module (pytest -s / -rA)


def _check_image(a, b, what="image"):
    d = np.abs(a - b)
    rmse = float(np.sqrt((d ** 2).mean()))
    assert rmse <= 1e-5, rmse
    n_out = int((d > 2e-5).sum())
    ALLOWANCE.append((what, n_out, int(d.size), float(d.max())))
    print(f"[allowance] {what}: {n_out} of {d.size} pixel values differ by more than 2e-5 (allowed {1e-4 * d.size:.0f}), max {d.max():.2e}")
    frac = n_out / d.size
    assert frac <= 1e-4, (frac, d.max())
    assert d.max() <= 1.0 / 255 + 1e-4, d.max()


def test_zzz_allowance_report():
    """Not a check of its own: prints how much of each tolerance allowance the tests of this module actually used (VERDICT
    r3: a regression INSIDE the allowance should be visible).  Runs last (name order)."""
    for what, n_out, n, worst in ALLOWANCE:
        print(f"[allowance] {what}: {n_out}/{n} outside the tight bound, worst {worst:.3e}")
    used = [a for a in ALLOWANCE if a[1] > 0]
    print(f"[allowance] {len(used)} of {len(ALLOWANCE)} comparisons used any of their allowance")


CASES = [
    # P, W, H, seed, extent, scale range
    (200, 64, 48, 0, 1.0, (0.005, 0.05)),
    (3000, 256, 256, 1, 1.0, (0.003, 0.04)),
    (20000, 200, 120, 2, 1.2, (0.002, 0.03)),      # ragged: 200x120 is not a multiple of 16
    (500, 97, 61, 3, 0.8, (0.02, 0.3)),            # big splats, odd image size
]


@pytest.mark.parametrize("P,W,H,seed,extent,srange", CASES)
def test_forward_matches_oracle(oracle32, P, W, H, seed, extent, srange):
    cam = look_at_camera((0.3, -3.0, 0.5), (0, 0, 0), W, H, fovx_deg=55.0)
    g = random_gaussians(P, seed=seed, extent=extent, scale_lo=srange[0], scale_hi=srange[1])
    bg = (0.1, 0.25, 0.4)
    ref = oracle32.render(cam.oracle_dict(bg=bg), **_kw(g))
    out = _run_gpu(cam, g, bg)
    assert (out["radii"] == ref["radii"]).all()
    _check_image(out["color"], ref["color"], f"forward P={P} {W}x{H}")


@pytest.mark.parametrize("P,W,H,seed,extent,srange", CASES[:3])
def test_backward_matches_oracle(oracle32, P, W, H, seed, extent, srange):
    cam = look_at_camera((0.3, -3.0, 0.5), (0, 0, 0), W, H, fovx_deg=55.0)
    g = random_gaussians(P, seed=seed, extent=extent, scale_lo=srange[0], scale_hi=srange[1])
    bg = (0.1, 0.25, 0.4)
    w = np.random.default_rng(seed).normal(size=(3, H, W)).astype(np.float32)
    ref = oracle32.render(cam.oracle_dict(bg=bg), **_kw(g), dL_dout=w)
    out = _run_gpu(cam, g, bg, w)
    _check_image(out["color"], ref["color"], f"backward-run image P={P} {W}x{H}")
    for k in ["dL_dmeans3D", "dL_dmeans2D", "dL_dcolors", "dL_dopacities", "dL_dscales", "dL_drotations"]:
        a, b = out[k], ref[k]
        scale = max(1e-6, float(np.abs(b).max()))
        err = np.abs(a - b) / scale
        # a flipped alpha-threshold decision changes one pixel's contribution; allow a tiny fraction of outliers
        n_out = int((err > 2e-4).sum())
        ALLOWANCE.append((f"{k} P={P}", n_out, int(err.size), float(err.max())))
        print(f"[allowance] {k} P={P}: {n_out} of {err.size} entries beyond 2e-4 of the maximum (allowed {2e-3 * err.size:.0f}), worst {err.max():.2e}")
        assert float((err > 2e-4).mean()) <= 2e-3, (k, float(err.max()), float((err > 2e-4).mean()))
        assert float(np.median(err)) <= 1e-6, (k, float(np.median(err)))


def test_scale_modifier_and_all_culled(oracle32):
    cam = look_at_camera((0.0, -3.0, 0.0), (0, 0, 0), 80, 64, fovx_deg=50.0)
    g = random_gaussians(300, seed=5)
    bg = (0.0, 0.0, 0.0)
    ref = oracle32.render(cam.oracle_dict(bg=bg, scale_modifier=0.5), **_kw(g))
    out = _run_gpu(cam, g, bg, scale_modifier=0.5)
    assert (out["radii"] == ref["radii"]).all()
    _check_image(out["color"], ref["color"])
    # everything behind the camera: empty lists, image == background, gradients zero
    g2 = random_gaussians(64, seed=6)
    g2["means3D"][:, 1] -= 20.0
    w = np.ones((3, 64, 80), dtype=np.float32)
    out = _run_gpu(cam, g2, (0.2, 0.4, 0.6), w)
    assert (out["radii"] == 0).all()
    assert np.allclose(out["color"][0], 0.2) and np.allclose(out["color"][2], 0.6)
    assert all(np.all(out[k] == 0) for k in out if k.startswith("dL_"))


def test_empty_input():
    from contextgs_amd.rasterizer import GaussianRasterizer
    cam = look_at_camera((0.0, -3.0, 0.0), (0, 0, 0), 32, 32)
    rast = GaussianRasterizer(_settings(cam, (0.5, 0.5, 0.5)))
    z = lambda *s: torch.zeros(*s, device="cuda")
    color, radii = rast(means3D=z(0, 3), means2D=z(0, 3), shs=None, colors_precomp=z(0, 3), opacities=z(0, 1),
                        scales=z(0, 3), rotations=z(0, 4), cov3D_precomp=None)
    assert radii.numel() == 0 and torch.allclose(color, torch.full_like(color, 0.5))
    assert rast.visible_filter(z(0, 3), z(0, 3), z(0, 4)).numel() == 0


def test_visible_filter_matches_oracle(oracle32):
    from contextgs_amd.rasterizer import GaussianRasterizer
    cam = orbit_cameras(4, 160, 90)[1]
    g = random_gaussians(50000, seed=7, extent=4.0)     # many outside the frustum / behind the camera
    ref = oracle32.visible_filter(cam.oracle_dict(), g["means3D"], g["scales"], g["rotations"])
    rast = GaussianRasterizer(_settings(cam, (0, 0, 0)))
    got = rast.visible_filter(*(torch.tensor(g[k], device="cuda") for k in ["means3D", "scales", "rotations"]))
    got = got.cpu().numpy()
    assert 0 < (ref > 0).sum() < ref.size
    assert (got == ref).all()


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 2047, 2048, 2049, 100_000, 5_000_017])
def test_scan_exact(n):
    import ctypes as C
    from contextgs_amd import _lib
    L = _lib.lib()
    x = torch.randint(0, 50, (max(n, 1),), device="cuda", dtype=torch.int32)[:n]
    out = torch.empty_like(x)
    scratch = torch.empty(L.cgs_scan_scratch_bytes(n), dtype=torch.uint8, device="cuda")
    _lib.check(L.cgs_scan_exclusive_u32(_lib.ptr(x), _lib.ptr(out), n, _lib.ptr(scratch), scratch.numel(),
                                        _lib.current_stream()), "scan")
    ref = torch.cumsum(x.to(torch.int64), 0) - x
    assert torch.equal(out.to(torch.int64), ref)


@pytest.mark.parametrize("n,lo,hi", [(1, 0, 32), (1000, 0, 32), (4096, 0, 8), (4097, 0, 13), (300_000, 0, 32),
                                     (2_000_003, 0, 13), (70_000, 4, 20), (6_000_001, 0, 32), (123_457, 3, 3)])
def test_radix_sort_stable_exact(n, lo, hi):
    from contextgs_amd import _lib
    L = _lib.lib()
    gen = torch.Generator(device="cuda").manual_seed(n)
    keys = torch.randint(0, 2 ** 31 - 1, (n,), device="cuda", dtype=torch.int64, generator=gen).to(torch.int32)
    if hi - lo <= 13:   # many duplicates: the stability test
        keys = keys & ((1 << hi) - 1)
    vals = torch.arange(n, device="cuda", dtype=torch.int32)
    ko, vo, kt, vt = (torch.empty_like(keys) for _ in range(4))
    scratch = torch.empty(L.cgs_sort_scratch_bytes(n), dtype=torch.uint8, device="cuda")
    _lib.check(L.cgs_sort_pairs_u32(_lib.ptr(keys), _lib.ptr(vals), _lib.ptr(ko), _lib.ptr(vo), _lib.ptr(kt),
                                    _lib.ptr(vt), n, lo, hi, _lib.ptr(scratch), scratch.numel(),
                                    _lib.current_stream()), "sort")
    digit = (keys.to(torch.int64) & 0xFFFFFFFF) >> lo & ((1 << (hi - lo)) - 1)
    order = torch.sort(digit, stable=True).indices
    assert torch.equal(vo.to(torch.int64), order)
    assert torch.equal(ko, keys[order])
    # vals_in == NULL: the values are the input positions (the depth sort's call), and a second sort on the same scratch
    # (the one-sweep path clears its own status words) gives the same answer
    for _ in range(2):
        vo.fill_(-1)
        _lib.check(L.cgs_sort_pairs_u32(_lib.ptr(keys), None, _lib.ptr(ko), _lib.ptr(vo), _lib.ptr(kt),
                                        _lib.ptr(vt), n, lo, hi, _lib.ptr(scratch), scratch.numel(),
                                        _lib.current_stream()), "sort")
        assert torch.equal(vo.to(torch.int64), order)
        assert torch.equal(ko, keys[order])


@pytest.mark.parametrize("n", [1, 4097, 300_000, 6_000_001])
def test_depth_key_sort_equals_the_32_bit_sort(n):
    """cgs_sort_depth_keys (27-bit keys, three 9-bit passes) == cgs_sort_pairs_u32 on the full float bits for depths inside the
    range, culled Gaussians (0xFFFFFFFF) anywhere behind the live ones; a depth beyond the range is REPORTED."""
    from contextgs_amd import _lib
    L = _lib.lib()
    gen = torch.Generator(device="cuda").manual_seed(n)
    # depths 0.2 .. ~12000, log-uniform, with exact duplicates (stability) and 5 % culled
    z = torch.exp(torch.empty(n, device="cuda").uniform_(math.log(0.2001), math.log(12000.0), generator=gen))
    z[1::2] = z[0::2][: n // 2].clone()
    keys = z.view(torch.int32).clone()
    culled = torch.rand(n, device="cuda", generator=gen) < 0.05
    keys[culled] = -1                                              # 0xFFFFFFFF
    ko, vo, kt, vt, ko2, vo2 = (torch.empty_like(keys) for _ in range(6))
    scratch = torch.empty(L.cgs_sort_scratch_bytes(n), dtype=torch.uint8, device="cuda")
    flag = torch.zeros(4, dtype=torch.int32, device="cuda")
    st = _lib.current_stream()
    _lib.check(L.cgs_sort_depth_keys(_lib.ptr(keys), _lib.ptr(ko), _lib.ptr(vo), _lib.ptr(kt), _lib.ptr(vt), n, _lib.ptr(scratch),
                                     scratch.numel(), _lib.ptr(flag), 77, st), "sort_depth_keys")
    _lib.check(L.cgs_sort_pairs_u32(_lib.ptr(keys), None, _lib.ptr(ko2), _lib.ptr(vo2), _lib.ptr(kt), _lib.ptr(vt), n, 0, 32,
                                    _lib.ptr(scratch), scratch.numel(), st), "sort")
    assert int(flag[0]) == 0
    live = int((~culled).sum())
    assert torch.equal(vo[:live], vo2[:live])                       # the live keys: the same order, entry for entry
    assert bool(culled[vo[live:].long()].all())                     # behind them only culled ones
    # one live depth beyond the range: reported with the caller's epoch, input keys untouched
    before = keys.clone()
    keys[n // 2] = torch.tensor([20000.0], device="cuda").view(torch.int32)[0]
    before[n // 2] = keys[n // 2]
    _lib.check(L.cgs_sort_depth_keys(_lib.ptr(keys), _lib.ptr(ko), _lib.ptr(vo), _lib.ptr(kt), _lib.ptr(vt), n, _lib.ptr(scratch),
                                     scratch.numel(), _lib.ptr(flag), 78, st), "sort_depth_keys")
    assert int(flag[0]) == 78 and torch.equal(keys, before)


def test_a_depth_beyond_the_27_bit_range_falls_back_to_the_32_bit_sort():
    """A view with live depths beyond ~13107: the preprocess wait sorts again on 32 bits, the speculative render is redone,
    the image equals the one of a thread that sorts on 32 bits from the start; later views of the thread stay on 32 bits."""
    from contextgs_amd import _lib
    from contextgs_amd.rasterizer import GaussianRasterizer
    L = _lib.lib()
    cam = look_at_camera((0.0, 0.0, -3.0), (0.0, 0.0, 0.0), 256, 256)
    g = random_gaussians(3000, seed=5, extent=1.0, scale_lo=0.01, scale_hi=0.05)
    t = {k: torch.tensor(v, device="cuda") for k, v in g.items()}
    # 40 large, far Gaussians straight ahead (z ~ 20000 .. 60000), overlapping on screen: their order matters
    far = 40
    t["means3D"][:far, :2] = torch.randn(far, 2, device="cuda") * 300.0
    t["means3D"][:far, 2] = torch.linspace(20000.0, 60000.0, far, device="cuda").flip(0)      # NOT in index order
    t["scales"][:far] = 1500.0
    t["opacities"][:far] = 0.6
    rast = GaussianRasterizer(_settings(cam, (0.1, 0.2, 0.3)))

    def run():
        return rast(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), shs=None, colors_precomp=t["colors"],
                    opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)[0]

    was = L.cgs_debug_set_depth_keys_full(1)
    try:
        ref = run()
        L.cgs_debug_set_depth_keys_full(0)
        run(); run()                                # (the pair capacity of this image size is learnt: the next view speculates)
        L.cgs_debug_set_depth_keys_full(0)
        out = run()                                 # ranged sort -> reported -> sorted again inside the wait, rendered again
        assert L.cgs_debug_set_depth_keys_full(0) == 1, "the thread did not switch to 32-bit depth keys"
        assert torch.equal(out, ref)
        assert float((ref - torch.tensor([0.1, 0.2, 0.3], device="cuda").view(3, 1, 1)).abs().max()) > 0.05
    finally:
        L.cgs_debug_set_depth_keys_full(was)


def test_full_hd_properties():
    """BASELINE-size image (1920x1080), many Gaussians: properties that do not need the oracle.
    linearity in the colours (render(c1)+render(c2) == render(c1+c2) with bg=0), bounded output,
    and determinism of the forward pass."""
    from contextgs_amd.rasterizer import GaussianRasterizer
    cam = orbit_cameras(8, 1920, 1080)[3]
    g = random_gaussians(400_000, seed=11, extent=1.0, scale_lo=0.001, scale_hi=0.01)
    t = {k: torch.tensor(v, device="cuda") for k, v in g.items()}
    rast = GaussianRasterizer(_settings(cam, (0.0, 0.0, 0.0)))

    def run(colors):
        return rast(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), shs=None, colors_precomp=colors,
                    opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)[0]

    c1 = t["colors"]
    c2 = torch.rand_like(c1)
    a, b, ab = run(c1), run(c2), run(c1 + c2)
    assert torch.allclose(a + b, ab, atol=2e-5)
    assert torch.equal(run(c1), a)
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 + 1e-5


@pytest.mark.parametrize("decoded", [False, True])
def test_prefilter_voxel_one_launch_equals_visible_filter(decoded):
    """renderer.prefilter_voxel (cgs_filter_voxel: exp of the three scale columns, the shared rotation and the `> 0`
    inside one kernel) against GaussianRasterizer.visible_filter on the materialised inputs
    (gaussian_renderer/__init__.py:262-287): the same bool mask, for a training model and a decoded one."""
    import torch
    from contextgs_amd.rasterizer import GaussianRasterizer
    from contextgs_amd.renderer import _raster_settings, prefilter_voxel
    from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
    pc = make_scene(60000, seed=4)
    if decoded:
        with torch.no_grad():
            pc._scaling.copy_(torch.exp(pc._scaling))
        pc.decoded_version = True
    pipe, bg = SynthPipe(), torch.zeros(3, device="cuda")
    for cam in orbit_cameras(3, 640, 360):
        cam = cam.to_torch("cuda")
        got = prefilter_voxel(cam, pc, pipe, bg)
        with torch.no_grad():
            sc = pc._scaling[:, :3] if decoded else torch.exp(pc._scaling[:, :3])
            rot0 = pc.rotation_activation(pc._rotation[:1])
            ref = GaussianRasterizer(_raster_settings(cam, pipe, bg, 1.0)).visible_filter(
                means3D=pc.get_anchor, scales=sc, rotations=rot0.repeat(sc.shape[0], 1), cov3D_precomp=None) > 0
        assert got.dtype == torch.bool and torch.equal(got, ref) and 0 < int(got.sum()) < got.numel()


def test_grid_above_65536_tiles_takes_the_32bit_key_path(oracle32):
    """4112 x 4112 pixels = 257 x 257 tiles: one more than the 16-bit tile keys of csrc/tile_bin.hip hold, so the
    rasterizer bins through emit_pairs + the 32-bit pair sort (csrc/api.hip); same image, same radii."""
    W = H = 4112
    cam = look_at_camera((0.3, -3.0, 0.5), (0, 0, 0), W, H, fovx_deg=55.0)
    g = random_gaussians(400, seed=11, extent=1.0, scale_lo=0.003, scale_hi=0.05)
    bg = (0.1, 0.25, 0.4)
    ref = oracle32.render(cam.oracle_dict(bg=bg), **_kw(g))
    out = _run_gpu(cam, g, bg)
    assert (out["radii"] == ref["radii"]).all()
    _check_image(out["color"], ref["color"])


def _bin_compare():
    """[differing list entries, differing range words] between the lists of the last forward and the round-1 binning"""
    import ctypes as C
    from contextgs_amd import _lib
    from contextgs_amd.rasterizer import last_call
    L = _lib.lib()
    R, R_ws, Pn = int(last_call["num_rendered"]), int(last_call["bin_R"]), int(last_call["P"])
    assert R > 0
    cfg = last_call["cfg"]
    dev = last_call["bin_ws"].device
    bin2 = torch.empty(int(L.cgs_raster_bin_bytes(Pn, R)), dtype=torch.uint8, device=dev)
    tiles = ((cfg.c.image_width + 15) // 16) * ((cfg.c.image_height + 15) // 16)
    ranges2 = torch.zeros(tiles * 2, dtype=torch.int32, device=dev)
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    f = L.cgs_debug_bin_compare
    _lib.check(f(cfg.ref, Pn, R, R_ws, _lib.ptr(last_call["geom_ws"]), last_call["geom_ws"].numel(),
                 _lib.ptr(last_call["bin_ws"]), last_call["bin_ws"].numel(), _lib.ptr(last_call["img_ws"]),
                 last_call["img_ws"].numel(), _lib.ptr(bin2), bin2.numel(), _lib.ptr(ranges2), _lib.ptr(out),
                 _lib.current_stream()), "cgs_debug_bin_compare")
    return out.tolist(), R, R_ws


@pytest.fixture
def bin_mode(request):
    """cgs_debug_set_bin_mode for one test: 1 = radix passes over (tile, Gaussian) pairs, 2 = two-level (bucket) binning"""
    from contextgs_amd import _lib
    L = _lib.lib()
    _lib.check(L.cgs_debug_set_bin_mode(request.param), "cgs_debug_set_bin_mode")
    yield request.param
    _lib.check(L.cgs_debug_set_bin_mode(0), "cgs_debug_set_bin_mode")


@pytest.mark.parametrize("bin_mode", [1, 2], indirect=True)
@pytest.mark.parametrize("P,W,H,scale_hi", [(4000, 256, 256, 0.05), (30000, 800, 800, 0.02), (20000, 1920, 1080, 0.08),
                                             (300, 97, 61, 0.4), (60000, 64, 64, 0.3), (50000, 1920, 1080, 0.5),
                                             (3000, 2600, 1800, 0.2), (70000, 1000, 700, 0.15), (9000, 333, 777, 0.6),
                                             (1025, 2048, 512, 0.9), (200000, 1280, 720, 0.01)])
def test_tile_lists_equal_the_pair_sort(P, W, H, scale_hi, bin_mode):
    """csrc/tile_bin.hip (pair-generating first radix pass, 16-bit tile keys, ranges from the last pass) leaves the same
    per-tile lists, entry for entry, and the same tile ranges as the round-1 binning (emit_pairs + stable 32-bit pair sort,
    the path the oracle comparisons of rounds 1-2 ran on): one pass (256 tiles), 6+6 and 7+6 bit passes, a 28-tile grid
    with splats that cover all of it, and 16 tiles with ~40 000 entries each, half of them exact depth ties (pairs of Gaussians
    at the same position: the ids decide).  Three renders of each scene: with the pair count known on the host, speculative
    (count read on the device, capacity from the first render), and speculative with a capacity that is too small
    (the view is rendered again with the true count).  Round 5: every scene under both binnings — the radix passes and the
    two-level path (bucket lists, then count + scan + fill; csrc/tile_bin.hip) — plus 50 000 screen-filling splats at 1080p
    (multi-chunk buckets, masks of all 32 tiles), a 163 x 113 tile grid (more than 256 buckets: mode 2 falls back), odd grids
    (63 x 44, 21 x 49 tiles: partial buckets on both edges), a 128 x 32 grid of exactly 128 buckets with splats across all of it,
    and 200 000 tiny splats (one or two tiles each: bucket lists of a few entries per chunk)."""
    from contextgs_amd import rasterizer as rz
    cam = look_at_camera((0.3, -3.0, 0.5), (0, 0, 0), W, H, fovx_deg=55.0)
    g = random_gaussians(P, seed=P, extent=1.0, scale_lo=0.003, scale_hi=scale_hi)
    if P == 60000:
        g["means3D"][1::2] = g["means3D"][0::2]
    rz._pair_capacity.pop((H, W), None)
    first = _run_gpu(cam, g, (0.0, 0.0, 0.0))["color"]
    diff, R, R_ws = _bin_compare()
    assert diff == [0, 0] and R_ws == R, (diff, R, R_ws)
    again = _run_gpu(cam, g, (0.0, 0.0, 0.0))["color"]              # speculative: the capacity comes from the first render
    diff, R2, R_ws = _bin_compare()
    assert diff == [0, 0] and R2 == R and R_ws == rz.pair_capacity_for(R), (diff, R2, R_ws)
    assert (again == first).all()
    rz._pair_capacity[(H, W)] = 4097                                 # too small for all but the 28-tile case
    third = _run_gpu(cam, g, (0.0, 0.0, 0.0))["color"]
    diff, R3, R_ws = _bin_compare()
    assert diff == [0, 0] and R3 == R and R_ws == (R if R > 4097 else 4097), (diff, R3, R_ws)
    assert R > 4097 or P == 300
    print(f"[binning] mode {bin_mode}: P {P} {W}x{H}: {R} pairs")
    assert rz._pair_capacity[(H, W)] == (rz.pair_capacity_for(R) if R > 4097 else 4097)
    assert (third == first).all()
