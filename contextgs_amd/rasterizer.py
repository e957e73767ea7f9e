"""Drop-in for the reference's `diff_gaussian_rasterization` package.

Same Python surface as the Scaffold-GS fork the reference imports
(gaussian_renderer/__init__.py:20): `GaussianRasterizationSettings`,
`GaussianRasterizer(raster_settings)(means3D, means2D, shs, colors_precomp,
opacities, scales, rotations, cov3D_precomp) -> (color[3,H,W], radii int32[P])`
and `.visible_filter(means3D, scales, rotations, cov3D_precomp) -> radii`.
All arithmetic runs in libcgs_hip.so (hand-written gfx950 kernels) through the
C-ABI of include/cgs.h; there is no CPU implementation here.

Only the argument combination the reference uses is implemented
(shs=None, colors_precomp given, scales+rotations given, cov3D_precomp=None;
gaussian_renderer/__init__.py:197-205, 280-285); the others raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """float32, contiguous.  A camera's matrices arrive as transposed views (scene/cameras.py builds them with .transpose(0, 1)):
    the dense copy is made once per tensor and kept on the tensor object, keyed by its version counter (two launches per step
    otherwise)."""
    if t.dtype == torch.float32 and t.is_contiguous():
        return t
    hit = getattr(t, "_cgs_f32c", None)
    if hit is not None and hit[0] == t._version:
        return hit[1]
    r = t.detach().float().contiguous()
    try:
        t._cgs_f32c = (t._version, r)
    except Exception:
        pass
    return r


class _Cfg:
    """Owns the ctypes struct plus the device tensors it points at."""

    def __init__(self, rs: GaussianRasterizationSettings):
        self.view = _f32c(rs.viewmatrix)
        self.proj = _f32c(rs.projmatrix)
        self.bg = _f32c(rs.bg)
        self.campos = _f32c(rs.campos) if rs.campos is not None else None
        _lib.require_device(self.view, self.proj, self.bg)
        self.c = _lib.RasterCfg(
            image_height=int(rs.image_height), image_width=int(rs.image_width),
            tanfovx=float(rs.tanfovx), tanfovy=float(rs.tanfovy),
            scale_modifier=float(rs.scale_modifier), prefiltered=int(bool(rs.prefiltered)),
            debug=int(bool(rs.debug)),
            viewmatrix=_lib.ptr(self.view), projmatrix=_lib.ptr(self.proj),
            campos=_lib.ptr(self.campos), bg=_lib.ptr(self.bg))

    @property
    def ref(self):
        return C.byref(self.c)


last_call: dict = {}   # sizes / image workspace of the most recent forward (bench.py's byte accounting)

# Pair-count speculation (cgs_raster_render_spec): the binning + blend of a view are enqueued before the host has read the
# view's pair count, into a workspace whose capacity is kept per image size (1.25 x the count that last exceeded it, rounded
# up to a whole 2^20 pairs, so that the workspace keeps ONE size and the caching allocator keeps handing out the same block:
# a capacity that crept up with every new maximum cost a fresh multi-GB hipMalloc each time); the count is
# read through an event while the device renders.  A view that needs more pairs than that is rendered again with its true
# count (same buffers, same stream: nothing has consumed the first attempt).  CGS_RASTER_SPEC=0 turns it off.
SPECULATE = os.environ.get("CGS_RASTER_SPEC", "1") != "0"
_pair_capacity: dict = {}     # (H, W) -> capacity (pairs) of the speculative binning workspace


def pair_capacity_for(num_rendered: int) -> int:
    """The capacity a view with `num_rendered` pairs sets for the following views of its image size."""
    return ((num_rendered + num_rendered // 4 + 4096 + (1 << 20) - 1) >> 20) << 20


def _workspace(nbytes: int, device) -> torch.Tensor:
    # (plain caching-allocator blocks.  A pool of size-classed buffers of our own was tried in round 5 while hunting 19 ms steps
    #  at 136 M pairs — it was not the cause (a leak in ctx_ops._LevelFused was) and cost 0.3 ms of host time per step: removed)
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def bin_and_blend(cfg, P, geom, img, color, stream, ticket):
    """Everything of a view's forward behind a cgs_raster_preprocess*_launch on `geom`: the speculative binning + blend with
    the image size's pair capacity, the read of the pair count, and the re-render with the true count when the capacity did
    not hold.  Returns (binning workspace, the pair count it was carved with = the backward's `R`, the view's pair count)."""
    L = _lib.lib()
    dev = geom.device
    H, W = cfg.c.image_height, cfg.c.image_width
    R = C.c_int64(0)
    tiles = ((H + 15) // 16) * ((W + 15) // 16)
    cap = _pair_capacity.get((H, W), 0) if (SPECULATE and P > 0 and tiles <= 65536) else 0
    binws = None
    if cap:
        binws = _workspace(L.cgs_raster_bin_bytes(P, cap), dev)
        _lib.check(L.cgs_raster_render_spec(cfg.ref, P, cap, _lib.ptr(geom), geom.numel(), _lib.ptr(binws),
                                            binws.numel(), _lib.ptr(img), img.numel(), _lib.ptr(color), stream),
                   "cgs_raster_render_spec")
    resorted = C.c_int(0)        # the view was sorted again on 32-bit depth keys (a depth beyond ~13107): the speculative render is void
    _lib.check(L.cgs_raster_preprocess_wait2(ticket, C.byref(R), C.byref(resorted)), "cgs_raster_preprocess_wait")
    num_rendered = int(R.value)
    if num_rendered > _pair_capacity.get((H, W), 0):
        _pair_capacity[(H, W)] = pair_capacity_for(num_rendered)
    bin_R = cap                              # the count the binning workspace was carved with (the backward's `R`)
    if not cap or num_rendered > cap or resorted.value:
        bin_R = num_rendered
        binws = _workspace(L.cgs_raster_bin_bytes(P, num_rendered), dev)
        _lib.check(L.cgs_raster_render(cfg.ref, P, num_rendered, _lib.ptr(geom), geom.numel(), _lib.ptr(binws),
                                       binws.numel(), _lib.ptr(img), img.numel(), _lib.ptr(color), stream),
                   "cgs_raster_render")
    last_call.update(P=P, num_rendered=num_rendered, bin_R=bin_R, img_ws=img, geom_ws=geom, bin_ws=binws, cfg=cfg)
    return binws, bin_R, num_rendered


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, colors, opacities, scales, rotations, raster_settings):
        L = _lib.lib()
        _lib.require_device(means3D, colors, opacities, scales, rotations)
        means3D_c, colors_c = _f32c(means3D), _f32c(colors)
        opac_c, scales_c, rots_c = _f32c(opacities), _f32c(scales), _f32c(rotations)
        P = means3D_c.shape[0]
        dev = means3D_c.device
        cfg = _Cfg(raster_settings)
        H, W = cfg.c.image_height, cfg.c.image_width
        stream = _lib.current_stream()
        ctx.set_materialize_grads(False)        # no zero tensors for the gradient of `radii`

        radii = torch.empty(P, dtype=torch.int32, device=dev)       # the preprocess kernel writes every entry (0 = culled)
        geom = _workspace(L.cgs_raster_geom_bytes(P), dev)
        img = _workspace(L.cgs_raster_img_bytes(H, W), dev)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        ticket = C.c_uint64(0)
        _lib.check(L.cgs_raster_preprocess_launch(cfg.ref, P, _lib.ptr(means3D_c), _lib.ptr(colors_c), _lib.ptr(opac_c),
                                                  _lib.ptr(scales_c), _lib.ptr(rots_c), _lib.ptr(geom), geom.numel(),
                                                  _lib.ptr(radii), stream, C.byref(ticket)), "cgs_raster_preprocess_launch")
        binws, bin_R, _num_rendered = bin_and_blend(cfg, P, geom, img, color, stream, ticket)
        ctx.cfg = cfg
        ctx.num_rendered = bin_R
        ctx.save_for_backward(means3D_c, colors_c, opac_c, scales_c, rots_c, radii, geom, binws, img)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_color, _grad_radii):
        L = _lib.lib()
        means3D, colors, opac, scales, rots, radii, geom, binws, img = ctx.saved_tensors
        cfg = ctx.cfg
        P = means3D.shape[0]
        dev = means3D.device
        if grad_color is None:
            return (None,) * 7
        g = _f32c(grad_color)
        # colour / opacity gradients are accumulated atomically by the blend backward: one zero fill for both; the other
        # four arrays are written for EVERY Gaussian by the preprocess backward (zeros for culled ones)
        acc = torch.zeros(P * 4, dtype=torch.float32, device=dev)
        d_colors, d_opac = acc[:3 * P].view(P, 3), acc[3 * P:].view(opac.shape)
        rest = torch.empty(P * (3 + 3 + 3 + 4), dtype=torch.float32, device=dev)
        parts = torch.split(rest, [3 * P, 3 * P, 3 * P, 4 * P])
        d_means3D, d_means2D = parts[0].view(P, 3), parts[1].view(P, 3)
        d_scales, d_rots = parts[2].view(P, 3), parts[3].view(P, 4)
        scratch = _workspace(L.cgs_raster_bwd_scratch_bytes(P), dev)
        _lib.check(L.cgs_raster_backward(
            cfg.ref, P, ctx.num_rendered, _lib.ptr(means3D), _lib.ptr(colors), _lib.ptr(opac), _lib.ptr(scales),
            _lib.ptr(rots), _lib.ptr(radii), _lib.ptr(geom), geom.numel(), _lib.ptr(binws), binws.numel(),
            _lib.ptr(img), img.numel(), _lib.ptr(g), _lib.ptr(d_means3D), _lib.ptr(d_means2D), _lib.ptr(d_colors),
            _lib.ptr(d_opac), _lib.ptr(d_scales), _lib.ptr(d_rots), _lib.ptr(scratch), scratch.numel(),
            _lib.current_stream()), "cgs_raster_backward")
        return d_means3D, d_means2D, d_colors, d_opac, d_scales, d_rots, None


def rasterize_gaussians(means3D, means2D, colors_precomp, opacities, scales, rotations, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, colors_precomp, opacities, scales, rotations, raster_settings)


def raster_stats(raster_settings: GaussianRasterizationSettings, img_ws: torch.Tensor) -> torch.Tensor:
    """[R_eff, non-empty tiles] of the last render that used `img_ws` (int64[2], device)."""
    L = _lib.lib()
    cfg = _Cfg(raster_settings)
    out = torch.zeros(2, dtype=torch.int64, device=img_ws.device)
    _lib.check(L.cgs_raster_stats(cfg.ref, _lib.ptr(img_ws), img_ws.numel(), _lib.ptr(out), _lib.current_stream()),
               "cgs_raster_stats")
    return out


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Near-plane frustum test (the only cull the rasterizer applies)."""
        with torch.no_grad():
            rs = self.raster_settings
            ones = torch.ones_like(positions[:, :1])
            z = (torch.cat([positions, ones], dim=1) @ rs.viewmatrix)[:, 2]
            return z > 0.2

    def visible_filter(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
        if cov3D_precomp is not None or scales is None or rotations is None:
            raise NotImplementedError("visible_filter: only the scales+rotations form is on the hot path "
                                      "(gaussian_renderer/__init__.py:280-285)")
        L = _lib.lib()
        _lib.require_device(means3D, scales, rotations)
        with torch.no_grad():
            m, s, r = _f32c(means3D), _f32c(scales), _f32c(rotations)
            N = m.shape[0]
            cfg = _Cfg(self.raster_settings)
            radii = torch.zeros(N, dtype=torch.int32, device=m.device)
            _lib.check(L.cgs_filter(cfg.ref, N, _lib.ptr(m), _lib.ptr(s), _lib.ptr(r), _lib.ptr(radii),
                                    _lib.current_stream()), "cgs_filter")
        return radii

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if shs is not None or colors_precomp is None:
            raise NotImplementedError("ContextGS renders with precomputed colours (shs=None, "
                                      "gaussian_renderer/__init__.py:200-201)")
        if cov3D_precomp is not None or scales is None or rotations is None:
            raise NotImplementedError("ContextGS passes scales+rotations (cov3D_precomp=None, "
                                      "gaussian_renderer/__init__.py:203-205)")
        return rasterize_gaussians(means3D, means2D, colors_precomp, opacities, scales, rotations,
                                   self.raster_settings)
