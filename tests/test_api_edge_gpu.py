"""Degenerate inputs of the entropy-model / quantiser / codec / kNN entry points (utils/entropy_models.py, utils/encodings.py,
simple_knn.distCUDA2 as the reference calls them): empty tensors, vanishing and huge scales, far-out values, one symbol,
a stream one symbol longer than a chunk, a million symbols over a wide range, fewer points than neighbours."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("n,scale,xval,Q", [(0, 1.0, 0.0, 1.0), (5, 0.0, 0.0, 1.0), (5, 1e-30, 0.0, 1.0), (5, 1.0, 1e6, 1.0),
                                           (5, 1e6, 0.0, 1.0), (5, 1.0, 0.0, 1e-9)])
def test_entropy_gaussian_stays_finite(n, scale, xval, Q):
    from contextgs_amd import entropy_models as em
    m = em.Entropy_gaussian(Q=1)
    x = torch.full((n, 4), xval, device=DEV, requires_grad=True)
    mean = torch.zeros(n, 4, device=DEV, requires_grad=True)
    sc = torch.full((n, 4), scale, device=DEV, requires_grad=True)
    bits = m(x, mean, sc, Q)
    assert bits.shape == (n, 4) and bool(torch.isfinite(bits).all()) and bool((bits >= 0).all())
    bits.sum().backward()
    assert all(bool(torch.isfinite(t.grad).all()) for t in (x, mean, sc))


def test_quantisers_on_empty_tensors():
    from contextgs_amd import encodings as enc
    assert enc.STE_multistep.apply(torch.zeros(0, 3, device=DEV), torch.ones(0, 1, device=DEV)).shape == (0, 3)
    assert enc.STE_binary.apply(torch.zeros(0, device=DEV)).shape == (0,)
    assert enc.Quantize_anchor.apply(torch.zeros(0, 3, device=DEV), torch.zeros(3, device=DEV), torch.ones(3, device=DEV))[0].shape == (0, 3)


@pytest.mark.parametrize("n,scale,Q,spread", [(0, 1.0, 1.0, 1.0), (1, 1.0, 1.0, 1.0), (5000, 1e-9, 1.0, 3.0), (5000, 1e3, 1.0, 3.0),
                                              (5000, 1.0, 1e-3, 0.01), (50001, 1.0, 1.0, 3.0), (1_000_000, 20.0, 1.0, 50.0),
                                              (4096, 1e-9, 1.0, 0.0)])
def test_gaussian_codec_round_trips_on_extreme_parameters(n, scale, Q, spread, tmp_path):
    from contextgs_amd import encodings as enc
    g = torch.Generator(device=DEV).manual_seed(n + 1)
    mean = torch.randn(n, device=DEV, generator=g) * spread
    sc = torch.full((n,), scale, device=DEV)
    q = torch.full((n,), Q, device=DEV)
    x = torch.round((mean + torch.randn(n, device=DEV, generator=g) * min(max(scale, 1e-3), 30.0)) / Q) * Q
    f = os.path.join(str(tmp_path), "s.b")
    _bytes, _nbits, mn, mx = enc.encoder_gaussian(x, mean, sc, q, file_name=f)
    y = enc.decoder_gaussian(mean, sc, q, file_name=f, min_value=mn, max_value=mx)
    assert torch.equal(x, y)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 64])
def test_knn_with_fewer_points_than_neighbours(n):
    from contextgs_amd import knn
    d = knn.distCUDA2(torch.randn(n, 3, device=DEV))
    assert d.shape == (n,) and not bool(torch.isnan(d).any())
    # the mean over three neighbours needs three neighbours: a missing one counts as FLT_MAX (so does the public simple_knn)
    assert bool((d < 1e30).all()) == (n >= 4)
    assert float(knn.distCUDA2(torch.zeros(10, 3, device=DEV)).abs().max()) == 0.0


def test_interleaved_count_launches_are_refused():
    """the library keeps ONE pinned count slot per kind and thread; an object that waits after another launch of its
    kind would read that launch's count.  renderer.ExpandCount / VisibleList refuse instead of mis-sizing their outputs."""
    from contextgs_amd.renderer import ExpandCount, VisibleList
    a = ExpandCount(torch.randn(100, 10, device="cuda"), torch.ones(100, 10, 1, device="cuda"), 10)
    b = ExpandCount(torch.randn(50, 10, device="cuda"), torch.ones(50, 10, 1, device="cuda"), 10)
    with pytest.raises(RuntimeError, match="interleave"):
        a.wait()
    assert 0 <= b.wait() <= 500 and b.wait() == b.P          # the owner still reads its own count (and caches it)
    m1 = torch.rand(1000, device="cuda") < 0.5
    v1, v2 = VisibleList(m1), VisibleList(~m1)
    with pytest.raises(RuntimeError, match="interleave"):
        v1.wait()
    assert torch.equal(v2.wait(), torch.nonzero(~m1)[:, 0]) and v2.wait() is v2.wait()


def test_launch_wait_tickets_are_checked_by_the_library():
    """the guard of the *_launch / *_wait pairs lives in the C ABI, not only in renderer.py — every launch hands out
    a ticket, a wait with a ticket that a later launch of the same kind has replaced fails with an error instead of returning that
    launch's count; so does a ticket of another kind, and 0."""
    import ctypes as C
    from contextgs_amd import _lib
    L = _lib.lib()
    dev = "cuda"
    m1 = (torch.rand(5000, device=dev) < 0.3).view(torch.uint8)
    m2 = (torch.rand(700, device=dev) < 0.5).view(torch.uint8)

    def launch(m):
        idx = torch.empty(m.numel(), dtype=torch.int64, device=dev)
        ws = torch.empty(int(L.cgs_nonzero_scratch_bytes(m.numel())), dtype=torch.uint8, device=dev)
        t = C.c_uint64(0)
        _lib.check(L.cgs_nonzero_launch(_lib.ptr(m), m.numel(), _lib.ptr(idx), _lib.ptr(ws), ws.numel(), _lib.current_stream(),
                                        C.byref(t)), "cgs_nonzero_launch")
        return t, idx, ws

    t1, _i1, _w1 = launch(m1)
    t2, i2, _w2 = launch(m2)
    assert t1.value != 0 and t2.value != 0 and t1.value != t2.value
    cnt = C.c_int64(-1)
    assert L.cgs_nonzero_wait(t1, C.byref(cnt)) != 0 and b"stale ticket" in L.cgs_last_error()
    assert L.cgs_nonzero_wait(C.c_uint64(0), C.byref(cnt)) != 0
    assert L.cgs_expand_count_wait(t2, C.byref(cnt)) != 0            # a nonzero ticket is not an expand_count ticket
    _lib.check(L.cgs_nonzero_wait(t2, C.byref(cnt)), "cgs_nonzero_wait")
    assert cnt.value == int(m2.sum()) and torch.equal(i2[:cnt.value], torch.nonzero(m2)[:, 0])
    _lib.check(L.cgs_nonzero_wait(t2, C.byref(cnt)), "cgs_nonzero_wait")      # waiting twice on the current ticket is fine
    assert cnt.value == int(m2.sum())
    assert L.cgs_nonzero_launch(_lib.ptr(m1), m1.numel(), None, None, 0, _lib.current_stream(), None) != 0     # NULL ticket


def test_old_wait_reports_a_void_speculative_render_after_a_depth_resort():
    """C ABI, round 6: a view with live depths beyond the 27-bit key range is sorted again on 32 bits inside the wait.  A caller
    that enqueued cgs_raster_render_spec between _launch and _wait must learn that its render ran on the first order:
    cgs_raster_preprocess_wait returns CGS_ERR_RESPEC (count valid), cgs_raster_preprocess_wait2 sets *order_changed; without a
    speculative render in between the old wait just returns CGS_OK."""
    import ctypes as C
    from contextgs_amd import _lib
    from contextgs_amd.rasterizer import _Cfg, GaussianRasterizationSettings
    from contextgs_amd.synth import look_at_camera, random_gaussians
    import math
    L = _lib.lib()
    cam = look_at_camera((0.0, 0.0, -3.0), (0.0, 0.0, 0.0), 128, 128).to_torch("cuda")
    g = random_gaussians(2000, seed=3, extent=1.0, scale_lo=0.01, scale_hi=0.05)
    t = {k: torch.tensor(v, device="cuda") for k, v in g.items()}
    t["means3D"][:20, :2] = 0.0
    t["means3D"][:20, 2] = torch.linspace(20000.0, 40000.0, 20, device="cuda")
    t["scales"][:20] = 1500.0
    rs = GaussianRasterizationSettings(
        image_height=128, image_width=128, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device="cuda"), scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=1, campos=cam.camera_center, prefiltered=False, debug=False)
    cfg = _Cfg(rs)
    P = 2000
    st = _lib.current_stream()

    def launch():
        radii = torch.empty(P, dtype=torch.int32, device="cuda")
        geom = torch.empty(int(L.cgs_raster_geom_bytes(P)), dtype=torch.uint8, device="cuda")
        tk = C.c_uint64(0)
        _lib.check(L.cgs_raster_preprocess_launch(cfg.ref, P, _lib.ptr(t["means3D"]), _lib.ptr(t["colors"]), _lib.ptr(t["opacities"]),
                                                  _lib.ptr(t["scales"]), _lib.ptr(t["rotations"]), _lib.ptr(geom), geom.numel(),
                                                  _lib.ptr(radii), st, C.byref(tk)), "launch")
        return tk, geom, radii

    def spec(geom):
        cap = 1 << 20
        binws = torch.empty(int(L.cgs_raster_bin_bytes(P, cap)), dtype=torch.uint8, device="cuda")
        img = torch.empty(int(L.cgs_raster_img_bytes(128, 128)), dtype=torch.uint8, device="cuda")
        color = torch.empty(3, 128, 128, device="cuda")
        _lib.check(L.cgs_raster_render_spec(cfg.ref, P, cap, _lib.ptr(geom), geom.numel(), _lib.ptr(binws), binws.numel(),
                                            _lib.ptr(img), img.numel(), _lib.ptr(color), st), "spec")
        return binws, img, color

    was = L.cgs_debug_set_depth_keys_full(0)
    try:
        R = C.c_int64(0)
        # (1) no speculative render in between: the old wait is fine, and the thread switched to 32-bit keys
        tk, geom, _r = launch()
        assert L.cgs_raster_preprocess_wait(tk, C.byref(R)) == 0 and R.value > 0
        assert L.cgs_debug_set_depth_keys_full(0) == 1
        # (2) with one: CGS_ERR_RESPEC = 5 from the old wait, the count still delivered
        tk, geom, _r = launch()
        keep = spec(geom)
        R2 = C.c_int64(0)
        assert L.cgs_raster_preprocess_wait(tk, C.byref(R2)) == 5 and b"cgs_raster_render" in L.cgs_last_error()
        assert R2.value == R.value
        L.cgs_debug_set_depth_keys_full(0)
        # (3) wait2 reports it instead
        tk, geom, _r = launch()
        keep = spec(geom)
        ch = C.c_int(0)
        _lib.check(L.cgs_raster_preprocess_wait2(tk, C.byref(R2), C.byref(ch)), "wait2")
        assert ch.value == 1 and R2.value == R.value
        del keep
    finally:
        L.cgs_debug_set_depth_keys_full(was)
        torch.cuda.synchronize()


@pytest.mark.parametrize("n,N,w", [(0, 10, 3), (1, 1, 1), (5000, 20000, 10), (149741, 1000000, 12), (1000, 1000, 256)])
def test_add_rows_equals_index_add_on_distinct_rows(n, N, w):
    from contextgs_amd import ctx_ops
    gen = torch.Generator(device="cuda").manual_seed(n + w)
    out = torch.randn(N, w, device="cuda", generator=gen)
    idx = torch.randperm(N, device="cuda", generator=gen)[:n].contiguous()
    g = torch.randn(n, w, device="cuda", generator=gen)
    ref = out.clone().index_add_(0, idx, g)
    assert torch.equal(ctx_ops.add_rows_(out, idx, g), ref)
