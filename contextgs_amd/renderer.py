"""Drop-in for the reference's `gaussian_renderer/__init__.py`:
`prefilter_voxel`, `generate_neural_gaussians`, `render` with the same
signatures, phase switches and return values (cited lines are that file).

What runs where: the visibility filter and the whole rasterizer are HIP
kernels (rasterizer.py); the anchor->Gaussian elementwise/compaction chain is
the fused HIP expansion (expand.hip) behind `_ExpandGaussians`; the three
anchor MLPs share their input and run as one fused fp32-MFMA launch each way
(mlp3.hip; rocprof showed the rocBLAS + element-wise version dominating the
step); the context model of the training / eval phases is context_model.py.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch
import torch.nn.functional as F

from . import _lib, anchor_gen, mlp
from .context_model import LazyRows, begin_step, gather_unique, multi_scale_generating, multi_scale_generating_visible
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

Q_FEAT, Q_SCALING, Q_OFFSETS = 1, 0.001, 0.2      # :40-42


def _c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


import threading as _threading

_OUTSTANDING = _threading.local()      # per thread and kind: the object whose *_launch owns the library's pinned count slot


def _own_slot(kind, obj):
    setattr(_OUTSTANDING, kind, obj)


def _check_slot(kind, obj):
    # The library keeps ONE pinned count slot per kind and thread (csrc: g_nz_slot, g_expand_slot): *_wait returns what the most
    # recent *_launch of the thread wrote.  An object that waits after another one of its kind was launched would read that
    # other object's count and mis-size its outputs — refuse instead.
    if getattr(_OUTSTANDING, kind, None) is not obj:
        raise RuntimeError(f"{type(obj).__name__}: another {kind} launch was issued on this thread before wait(); "
                           "launch / wait pairs of one kind must not interleave")


class VisibleList:
    """torch.nonzero(mask)[:, 0] in two halves (cgs_nonzero_launch / _wait): the kernels are enqueued at construction,
    `wait()` blocks only until the count has been copied back and returns the index tensor."""

    def __init__(self, mask):
        L = _lib.lib()
        m = mask if mask.dtype == torch.uint8 else mask.view(torch.uint8)
        self.mask = m = m.contiguous().reshape(-1)
        n = int(m.numel())
        self.idx = torch.empty(n, dtype=torch.int64, device=m.device)
        self.scratch = torch.empty(int(L.cgs_nonzero_scratch_bytes(n)), dtype=torch.uint8, device=m.device)
        self.ticket = C.c_uint64(0)      # the library refuses a wait whose ticket a later launch of the kind has replaced
        _lib.check(L.cgs_nonzero_launch(_lib.ptr(m), n, _lib.ptr(self.idx), _lib.ptr(self.scratch), self.scratch.numel(),
                                        _lib.current_stream(), C.byref(self.ticket)), "cgs_nonzero_launch")
        self._idx = None
        _own_slot("nonzero", self)

    def wait(self):
        if self._idx is not None:
            return self._idx
        _check_slot("nonzero", self)
        cnt = C.c_int64(0)
        _lib.check(_lib.lib().cgs_nonzero_wait(self.ticket, C.byref(cnt)), "cgs_nonzero_wait")
        idx = self.idx[:int(cnt.value)]
        idx._cgs_ascending = True        # row gathers by this list take the one-pass backward (cgs_scatter_rows_sorted)
        self._idx = idx
        return idx


class ExpandCount:
    """Pass A of the expansion (mask * opacity, survivor flags, their scan) enqueued WITHOUT reading the survivor count:
    `launch` returns at once, the caller may enqueue independent work (the rate model of the step), `wait` then blocks
    only until the count has arrived — the device does not drain at this read-back."""

    def __init__(self, op_raw, masks, K):
        L = _lib.lib()
        op_raw, masks = _c(op_raw.detach()), _c(masks.detach())
        _lib.require_device(op_raw, masks)
        n = int(op_raw.shape[0])
        dev = op_raw.device
        slots = n * K
        self.n, self.K = n, K
        self.neural_opacity = torch.empty(slots, 1, dtype=torch.float32, device=dev)
        self.mask_out = torch.empty(slots, dtype=torch.bool, device=dev)
        self.flags = self.mask_out        # (the survivor flags ARE the bytes of mask_out: the kernels scan and read those)
        self.pos = torch.empty(slots, dtype=torch.int32, device=dev)
        self.scratch = torch.empty(L.cgs_expand_scratch_bytes(n, K), dtype=torch.uint8, device=dev)
        self.P = None
        self.ticket = C.c_uint64(0)
        _lib.check(L.cgs_expand_count_launch(n, K, _lib.ptr(op_raw), _lib.ptr(masks), _lib.ptr(self.neural_opacity),
                                             _lib.ptr(self.mask_out), None, _lib.ptr(self.pos),
                                             _lib.ptr(self.scratch), self.scratch.numel(), _lib.current_stream(),
                                             C.byref(self.ticket)), "cgs_expand_count_launch")
        _own_slot("expand_count", self)

    def wait(self) -> int:
        if self.P is None:
            _check_slot("expand_count", self)
            cnt = C.c_int64(0)
            _lib.check(_lib.lib().cgs_expand_count_wait(self.ticket, C.byref(cnt)), "cgs_expand_count_wait")
            self.P = int(cnt.value)
        return self.P


class _ExpandGaussians(torch.autograd.Function):
    """(anchor, grid_scaling, offsets, masks, mlp outputs) -> compacted Gaussians (:112-145)."""

    @staticmethod
    def forward(ctx, anchor, gscaling, offsets, masks, op_raw, color_in, cov_in, K, src_row=None, pre=None):
        # src_row: gscaling / offsets are the context model's full coding-order outputs and visible anchor n uses
        # their row src_row[n] (the visibility gather is done by the kernel)
        # pre: an ExpandCount already launched for (op_raw, masks) by the caller
        L = _lib.lib()
        _lib.require_device(anchor, gscaling, offsets, masks, op_raw, color_in, cov_in)
        anchor, gscaling, offsets = _c(anchor), _c(gscaling), _c(offsets)
        masks, op_raw, color_in, cov_in = _c(masks), _c(op_raw), _c(color_in), _c(cov_in)
        n = anchor.shape[0]
        dev = anchor.device
        stream = _lib.current_stream()
        if pre is None:
            pre = ExpandCount(op_raw, masks, K)
        assert pre.n == n and pre.K == K
        ctx.set_materialize_grads(False)        # an unused output (neural_opacity in most losses) arrives as None, not as zeros
        P = pre.wait()
        neural_opacity, mask_out, flags, pos = pre.neural_opacity, pre.mask_out, pre.flags, pre.pos
        xyz = torch.empty(P, 3, dtype=torch.float32, device=dev)
        color = torch.empty(P, 3, dtype=torch.float32, device=dev)
        opacity = torch.empty(P, 1, dtype=torch.float32, device=dev)
        scaling = torch.empty(P, 3, dtype=torch.float32, device=dev)
        rot = torch.empty(P, 4, dtype=torch.float32, device=dev)
        _lib.check(L.cgs_expand_write(n, K, _lib.ptr(flags), _lib.ptr(pos), _lib.ptr(anchor), _lib.ptr(gscaling),
                                      _lib.ptr(offsets), _lib.ptr(neural_opacity), _lib.ptr(color_in),
                                      _lib.ptr(cov_in), _lib.ptr(xyz), _lib.ptr(color), _lib.ptr(opacity),
                                      _lib.ptr(scaling), _lib.ptr(rot), _lib.ptr(src_row), stream), "cgs_expand_write")
        ctx.K = K
        ctx.n = n
        ctx.P = P                   # the substitutes of absent output gradients are sized from this, not from g_xyz
        ctx.src_row = src_row
        ctx.save_for_backward(flags, pos, gscaling, offsets, op_raw, masks, cov_in)
        ctx.mark_non_differentiable(mask_out)
        return xyz, color, opacity, scaling, rot, neural_opacity, mask_out

    @staticmethod
    def backward(ctx, g_xyz, g_color, g_opacity, g_scaling, g_rot, g_no, _g_mask):
        L = _lib.lib()
        flags, pos, gscaling, offsets, op_raw, masks, cov_in = ctx.saved_tensors
        K = ctx.K
        n = ctx.n
        src_row = ctx.src_row
        dev = gscaling.device
        P = ctx.P
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        g_xyz = _c(g_xyz) if g_xyz is not None else z(P, 3)
        g_color = _c(g_color) if g_color is not None else z(P, 3)
        g_opacity = _c(g_opacity) if g_opacity is not None else z(P, 1)
        g_scaling = _c(g_scaling) if g_scaling is not None else z(P, 3)
        g_rot = _c(g_rot) if g_rot is not None else z(P, 4)
        g_no = _c(g_no) if g_no is not None else None
        e = lambda t: torch.empty_like(t)
        d_anchor = torch.empty(n, 3, dtype=torch.float32, device=dev)
        d_op, d_mask = e(op_raw), e(masks)
        if src_row is None:
            d_gs, d_off = e(gscaling), e(offsets)
        else:       # rows of the larger arrays that no visible anchor reads get a zero gradient (zero_unlisted_rows: only those
            # rows are written); when every anchor is visible (src_row names every row) there is nothing to zero
            flat = torch.empty(gscaling.numel() + offsets.numel(), dtype=torch.float32, device=dev)
            d_gs, d_off = flat[:gscaling.numel()].view_as(gscaling), flat[gscaling.numel():].view_as(offsets)
            if n != gscaling.shape[0]:
                from .ctx_ops import zero_unlisted_rows
                zero_unlisted_rows(src_row, gscaling.shape[0], [d_gs, d_off])
        d_color = torch.empty(n, 3 * K, dtype=torch.float32, device=dev)
        d_cov = e(cov_in)
        _lib.check(L.cgs_expand_backward(
            n, K, _lib.ptr(flags), _lib.ptr(pos), _lib.ptr(gscaling), _lib.ptr(offsets), _lib.ptr(op_raw),
            _lib.ptr(masks), _lib.ptr(cov_in), _lib.ptr(g_xyz), _lib.ptr(g_color), _lib.ptr(g_opacity),
            _lib.ptr(g_scaling), _lib.ptr(g_rot), _lib.ptr(g_no), _lib.ptr(d_anchor), _lib.ptr(d_gs),
            _lib.ptr(d_off), _lib.ptr(d_op), _lib.ptr(d_mask), _lib.ptr(d_color), _lib.ptr(d_cov),
            _lib.ptr(src_row), _lib.current_stream()), "cgs_expand_backward")
        return d_anchor, d_gs, d_off, d_mask, d_op, d_color, d_cov, None, None, None


EARLY_MLP3 = os.environ.get("CGS_EARLY_MLP3", "1") != "0"            # A/B knob: 0 = the anchor MLPs are enqueued where their node is created
MASK_PAIR_NODE = os.environ.get("CGS_MASK_PAIR_NODE", "1") != "0"    # A/B knob: 0 = the mask weights' two row gathers as two nodes (round 5)
FUSE_VIEW = os.environ.get("CGS_FUSE_VIEW", "1") != "0"    # tuning / A-B knob: expansion fused with the rasterizer's preprocess


class ViewFusion:
    """What render() hands to generate_neural_gaussians when the expansion may run fused with the rasterizer: the view's
    raster settings in, the rendered view out (`done`: (image, radii, screenspace_points))."""

    def __init__(self, raster_settings, retain_grad):
        self.raster_settings, self.retain_grad, self.done = raster_settings, retain_grad, None


class _ExpandRasterize(torch.autograd.Function):
    """_ExpandGaussians followed by the rasterizer (rasterizer._RasterizeGaussians) as ONE node whose forward goes from the
    expansion's slots to the rasterizer's records in one kernel (csrc/expand_raster.hip): colour and opacity never exist as
    per-Gaussian tensors and nothing the expansion writes is read back by the preprocess stage.  The backward is the two
    nodes' backward launches (cgs_raster_backward, cgs_expand_backward) with their hand-over buffers kept inside the node.
    Same device functions on the same values as the two nodes: bit-identical image, radii and scaling
    (tests/test_fused_view_gpu.py).  means2D: the view's screenspace_points (its gradient is the only thing it
    carries), created by the caller once the survivor count P is known."""

    @staticmethod
    def forward(ctx, anchor, gscaling, offsets, masks, op_raw, color_in, cov_in, means2D, K, src_row, pre, raster_settings):
        from . import rasterizer as rz
        L = _lib.lib()
        _lib.require_device(anchor, gscaling, offsets, masks, op_raw, color_in, cov_in)
        anchor, gscaling, offsets = _c(anchor), _c(gscaling), _c(offsets)
        masks, op_raw, color_in, cov_in = _c(masks), _c(op_raw), _c(color_in), _c(cov_in)
        n = anchor.shape[0]
        dev = anchor.device
        stream = _lib.current_stream()
        assert pre.n == n and pre.K == K
        ctx.set_materialize_grads(False)
        P = pre.wait()
        assert means2D.shape[0] == P
        neural_opacity, mask_out, flags, pos = pre.neural_opacity, pre.mask_out, pre.flags, pre.pos
        cfg = rz._Cfg(raster_settings)
        H, W = cfg.c.image_height, cfg.c.image_width
        flat = torch.empty(P * 10, dtype=torch.float32, device=dev)          # scaling | xyz | rot (kept for the backward)
        scaling, xyz, rot = flat[:3 * P].view(P, 3), flat[3 * P:6 * P].view(P, 3), flat[6 * P:].view(P, 4)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        geom = rz._workspace(L.cgs_raster_geom_bytes(P), dev)
        img = rz._workspace(L.cgs_raster_img_bytes(H, W), dev)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        ticket = C.c_uint64(0)
        _lib.check(L.cgs_raster_preprocess_expand_launch(
            cfg.ref, n, K, _lib.ptr(flags), _lib.ptr(pos), _lib.ptr(anchor), _lib.ptr(gscaling), _lib.ptr(offsets),
            _lib.ptr(neural_opacity), _lib.ptr(color_in), _lib.ptr(cov_in), _lib.ptr(src_row), P, _lib.ptr(scaling),
            _lib.ptr(xyz), _lib.ptr(rot), _lib.ptr(geom), geom.numel(), _lib.ptr(radii), stream, C.byref(ticket)),
            "cgs_raster_preprocess_expand_launch")
        binws, bin_R, _num_rendered = rz.bin_and_blend(cfg, P, geom, img, color, stream, ticket)
        ctx.cfg, ctx.num_rendered, ctx.K, ctx.n, ctx.src_row, ctx.P = cfg, bin_R, K, n, src_row, P
        ctx.save_for_backward(flags, pos, gscaling, offsets, op_raw, masks, cov_in, xyz, scaling, rot, radii, geom, binws, img)
        ctx.mark_non_differentiable(radii, mask_out)
        return color, radii, scaling, neural_opacity, mask_out

    @staticmethod
    def backward(ctx, g_img, _g_radii, g_scaling, g_no, _g_mask):
        from . import rasterizer as rz
        L = _lib.lib()
        (flags, pos, gscaling, offsets, op_raw, masks, cov_in, xyz, scaling, rot, radii, geom, binws, img) = ctx.saved_tensors
        cfg, K, n, src_row, P = ctx.cfg, ctx.K, ctx.n, ctx.src_row, ctx.P
        dev = gscaling.device
        stream = _lib.current_stream()
        # ---- the rasterizer's backward (as _RasterizeGaussians.backward) into buffers that stay inside this node
        acc = torch.zeros(max(P, 1) * 4, dtype=torch.float32, device=dev)      # atomically accumulated: dL/d colour | opacity
        d_colors, d_opac = acc[:3 * P].view(P, 3), acc[3 * P:4 * P].view(P, 1)
        rest = torch.empty(max(P, 1) * 13, dtype=torch.float32, device=dev)
        d_means3D, d_means2D = rest[:3 * P].view(P, 3), rest[3 * P:6 * P].view(P, 3)
        d_scales, d_rots = rest[6 * P:9 * P].view(P, 3), rest[9 * P:13 * P].view(P, 4)
        if g_img is not None and P > 0:
            scratch = rz._workspace(L.cgs_raster_bwd_scratch_bytes(P), dev)
            _lib.check(L.cgs_raster_backward(
                cfg.ref, P, ctx.num_rendered, _lib.ptr(xyz), None, None, _lib.ptr(scaling), _lib.ptr(rot), _lib.ptr(radii),
                _lib.ptr(geom), geom.numel(), _lib.ptr(binws), binws.numel() if binws is not None else 0, _lib.ptr(img),
                img.numel(), _lib.ptr(rz._f32c(g_img)), _lib.ptr(d_means3D), _lib.ptr(d_means2D), _lib.ptr(d_colors),
                _lib.ptr(d_opac), _lib.ptr(d_scales), _lib.ptr(d_rots), _lib.ptr(scratch), scratch.numel(), stream),
                "cgs_raster_backward")
        else:
            rest.zero_()
        if g_scaling is not None:            # the loss reads `scaling` too (train.py:204): what autograd would have summed
            d_scales = d_scales + _c(g_scaling)
        # ---- the expansion's backward (as _ExpandGaussians.backward)
        g_no = _c(g_no) if g_no is not None else None
        e = lambda t: torch.empty_like(t)
        d_anchor = torch.empty(n, 3, dtype=torch.float32, device=dev)
        d_op, d_mask = e(op_raw), e(masks)
        if src_row is None:
            d_gs, d_off = e(gscaling), e(offsets)
        else:       # rows of the larger arrays that no visible anchor reads get zeros (ctx_ops.zero_unlisted_rows: only
            #             THOSE rows are written, not a 144 MB fill); every row read -> nothing to zero
            flat = torch.empty(gscaling.numel() + offsets.numel(), dtype=torch.float32, device=dev)
            d_gs, d_off = flat[:gscaling.numel()].view_as(gscaling), flat[gscaling.numel():].view_as(offsets)
            if n != gscaling.shape[0]:
                from .ctx_ops import zero_unlisted_rows
                zero_unlisted_rows(src_row, gscaling.shape[0], [d_gs, d_off])
        d_color = torch.empty(n, 3 * K, dtype=torch.float32, device=dev)
        d_cov = e(cov_in)
        _lib.check(L.cgs_expand_backward(
            n, K, _lib.ptr(flags), _lib.ptr(pos), _lib.ptr(gscaling), _lib.ptr(offsets), _lib.ptr(op_raw), _lib.ptr(masks),
            _lib.ptr(cov_in), _lib.ptr(d_means3D), _lib.ptr(d_colors), _lib.ptr(d_opac), _lib.ptr(d_scales), _lib.ptr(d_rots),
            _lib.ptr(g_no), _lib.ptr(d_anchor), _lib.ptr(d_gs), _lib.ptr(d_off), _lib.ptr(d_op), _lib.ptr(d_mask),
            _lib.ptr(d_color), _lib.ptr(d_cov), _lib.ptr(src_row), stream), "cgs_expand_backward")
        return d_anchor, d_gs, d_off, d_mask, d_op, d_color, d_cov, d_means2D, None, None, None, None


def _anchor_mlps(pc, x):
    """mlp_opacity / mlp_color / mlp_cov on the shared [N,54] input (:112,122,126).
    The three first layers are one GEMM; outputs are identical dot products."""
    mo, mc, mv = pc.get_opacity_mlp, pc.get_color_mlp, pc.get_cov_mlp
    if mlp.anchor_mlp3_supported(mo, mc, mv):
        # the three MLPs share x: one fused launch forward, one backward (csrc/mlp3.hip)
        return mlp.anchor_mlp3(x, mo, mc, mv)
    if mlp.supported(mo) and mlp.supported(mc) and mlp.supported(mv):
        # one fused fp32-MFMA launch per MLP forward, three backward (csrc/mlp.hip): rocprof showed the
        # rocBLAS + elementwise + bias-reduction version of these skinny MLPs dominating the step
        return mlp.mlp2(x, mo), mlp.mlp2(x, mc), mlp.mlp2(x, mv)
    D = mo[0].out_features
    W1 = torch.cat([mo[0].weight, mc[0].weight, mv[0].weight], dim=0)
    b1 = torch.cat([mo[0].bias, mc[0].bias, mv[0].bias], dim=0)
    h = F.relu(F.linear(x, W1, b1))
    op_raw = torch.tanh(F.linear(h[:, :D], mo[2].weight, mo[2].bias))
    color = torch.sigmoid(F.linear(h[:, D:2 * D], mc[2].weight, mc[2].bias))
    cov = F.linear(h[:, 2 * D:], mv[2].weight, mv[2].bias)
    return op_raw, color, cov


def _generate_fused(viewpoint_camera, pc, anchor, feat, grid_scaling, grid_offsets, binary_grid_masks, K):
    """:106-145 as one autograd node over the fused kernel family (anchor_gen.py): MLP input assembly, the three anchor
    MLPs, mask, compaction and the per-Gaussian tail.  None when the shapes are not the ones the kernels are built for."""
    mo, mc, mv = pc.get_opacity_mlp, pc.get_color_mlp, pc.get_cov_mlp
    fs, fr = (feat.src, feat.idx) if isinstance(feat, LazyRows) else (feat, None)
    lazy_s, lazy_o = isinstance(grid_scaling, LazyRows), isinstance(grid_offsets, LazyRows)
    if lazy_s and lazy_o and grid_scaling.idx is grid_offsets.idx:
        gsrc, osrc, grow = grid_scaling.src, grid_offsets.src, grid_scaling.idx
    elif not lazy_s and not lazy_o:
        gsrc, osrc, grow = grid_scaling, grid_offsets, None
    else:
        return None
    if not (fs.dim() == 2 and fs.shape[1] == 50 and gsrc.dim() == 2 and gsrc.shape[1] == 6
            and anchor_gen.supported(mo, mc, mv, K, fs, fr, anchor, gsrc, osrc, grow, binary_grid_masks)):
        return None
    return anchor_gen.anchor_gen(fs, fr, anchor, viewpoint_camera.camera_center, gsrc, osrc, grow,
                                 binary_grid_masks.reshape(-1, K), mo, mc, mv)


def _generate_unfused(viewpoint_camera, pc, anchor, feat, grid_scaling, grid_offsets, binary_grid_masks, K, between=None,
                      view=None, mlp_pre=None):
    """The same stage as separate launches: fused three-MLP kernel (or torch) + the expansion kernels.
    between: called after the expansion's survivor count has been ENQUEUED and before it is read — work it launches
    (the step's rate model) runs on the device while the host waits for the count."""
    mo, mc, mv = pc.get_opacity_mlp, pc.get_color_mlp, pc.get_cov_mlp
    if (isinstance(feat, LazyRows) and feat.src.dim() == 2 and feat.src.shape[1] == 50 and feat.src.is_cuda
            and mlp.anchor_mlp3_supported(mo, mc, mv)):
        # :106-127 in one launch: the visibility gather of the context model's output, the view direction / distance
        # and the [feat, view, dist] concatenation happen inside the fused three-MLP kernel's operand load (and its
        # backward scatters straight into the source rows): no [n,54] gather / scatter / norm / div launches
        op_raw, color_in, cov_in = mlp.anchor_mlp3_rows(feat.src, feat.idx, anchor, viewpoint_camera.camera_center,
                                                        mo, mc, mv, pre=mlp_pre)
    else:
        ob_view = anchor - viewpoint_camera.camera_center                               # :106-110
        ob_dist = ob_view.norm(dim=1, keepdim=True)
        ob_view = ob_view / ob_dist
        if isinstance(feat, LazyRows):
            # the visibility gather of the context model's output and this concatenation are one launch (and one
            # scatter on the way back) instead of a gather + a cat
            cat_local_view = feat.cat_with(ob_view, ob_dist)
        else:
            cat_local_view = torch.cat([feat, ob_view, ob_dist], dim=1)
        op_raw, color_in, cov_in = _anchor_mlps(pc, cat_local_view)                      # :112-127
    src_row = None
    if isinstance(grid_scaling, LazyRows) and isinstance(grid_offsets, LazyRows) and grid_scaling.idx is grid_offsets.idx:
        # the expansion kernels read the context model's coding-order outputs through the row index themselves
        src_row, grid_scaling, grid_offsets = grid_scaling.idx, grid_scaling.src, grid_offsets.src
    else:
        grid_scaling = grid_scaling.materialize() if isinstance(grid_scaling, LazyRows) else grid_scaling
        grid_offsets = grid_offsets.materialize() if isinstance(grid_offsets, LazyRows) else grid_offsets
    masks2 = binary_grid_masks.reshape(-1, K)
    pre = None
    if op_raw.is_cuda and masks2.is_cuda:
        pre = ExpandCount(op_raw, masks2, K)
    if between is not None:
        between()
    if (view is not None and FUSE_VIEW and pre is not None and anchor.is_cuda and grid_scaling.is_cuda
            and grid_offsets.is_cuda and torch.is_grad_enabled()):
        # the expansion and the rasterizer as one node (csrc/expand_raster.hip); the survivor count is read here, after
        # `between` has put the rate model in the queue, so that screenspace_points can be an INPUT of the node
        P = pre.wait()
        screenspace_points = _zero_points(torch.empty(P, 0, device=anchor.device))
        if view.retain_grad:
            try:
                screenspace_points.retain_grad()
            except Exception:
                pass
        image, radii, scaling, neural_opacity, mask = _ExpandRasterize.apply(
            anchor, grid_scaling, grid_offsets, masks2, op_raw, color_in, cov_in, screenspace_points, K, src_row, pre,
            view.raster_settings)
        view.done = (image, radii, screenspace_points)
        return None, None, None, scaling, None, neural_opacity, mask
    return _ExpandGaussians.apply(anchor, grid_scaling, grid_offsets, masks2, op_raw, color_in, cov_in, K, src_row, pre)


def generate_neural_gaussians(viewpoint_camera, pc, visible_mask=None, is_training=False, step=0, _view=None):   # :25-150
    time_sub = 0
    if visible_mask is None:
        visible_mask = torch.ones(pc.get_anchor.shape[0], dtype=torch.bool, device=pc.get_anchor.device)

    bit_per_param = bit_per_anchor_param = bit_per_feat_param = None
    bit_per_scaling_param = bit_per_offsets_param = bpp_per_level = None

    # the visible-anchor list is ENQUEUED first and read after everything below that does not need it (:44-50)
    vis_pending = VisibleList(visible_mask) if visible_mask.is_cuda else None
    full_anchor = pc.get_anchor                      # one Quantize_anchor launch per call (the reference re-derives
    use_context = (is_training and step > 10000) or (not is_training and not pc.decoded_version)   # it ~5x)
    begun = binary_all = mask_anchor_bool = rate_thunk = late_mask = scaling_all = None
    if use_context:
        # everything of the context model that does not depend on the visible set is enqueued BEFORE the read-back of
        # the visible count below (the accessors, the step's bookkeeping kernel): the GPU works through it while the
        # host waits, and the context model's own count read-back finds its data already there
        late_mask = None
        if (is_training and hasattr(pc, "get_mask_pair") and not pc.decoded_version and pc._mask.is_cuda and pc._mask.requires_grad
                and torch.is_grad_enabled()):
            # the mask VALUES now (the level plan needs the anchor mask), its autograd nodes after the level loop's
            # (ctx_ops._MaskSTEAttach: `_mask`'s gradient is then final BEFORE the level kernels run in the backward)
            from . import ctx_ops as _ctx_ops
            binary_vals, mask_anchor_bool = _ctx_ops.mask_ste_values(pc._mask)
            late_mask = {"vals": binary_vals, "node": None}

            def binary_all():
                if late_mask["node"] is None:
                    late_mask["node"] = _ctx_ops.mask_ste_attach(pc._mask, late_mask["vals"])
                return late_mask["node"]
        elif hasattr(pc, "get_mask_pair"):
            binary_all, mask_anchor_bool = pc.get_mask_pair()
        else:                                   # the reference's own GaussianModel
            binary_all = pc.get_mask
            mask_anchor_bool = binary_all.detach().sum(dim=1)[:, 0] > 0
        begun = begin_step(pc, full_anchor, mask_anchor_bool, is_training)
        if begun is not None and is_training:
            # the level kernels of the step's context model depend on nothing the host still has to read (not on the view either):
            # they go into the queue here, 0.4 ms of device work behind which the read-backs below and the small launches between
            # them disappear (context_model._early_levels)
            from .context_model import early_levels_begin
            scaling_all = pc.get_scaling
            early_levels_begin(pc, full_anchor, pc._hyper_latent, pc._anchor_feat, pc._offset, scaling_all, mask_anchor_bool, begun)
    # visible rows are distinct anchors: gather by index with a sort-free scatter backward (the reference's
    # boolean-mask indexing, :44-50, backpropagates through index_put_(accumulate) = a device sort per tensor)
    vis_idx = vis_pending.wait() if vis_pending is not None else torch.nonzero(visible_mask)[:, 0]
    if is_training:
        # multi-GPU: the per-anchor gradients of this view only reach the visible anchors unless the context model runs over all
        # of them (dist.GradientSync(sparse="auto") exchanges the union of the ranks' rows instead of dense tensors)
        from . import dist as _dist
        _dist.note_touched_rows(None if use_context else visible_mask, int(vis_idx.shape[0]))
    sel = lambda t: gather_unique(t, vis_idx)
    anchor = sel(full_anchor)
    early = begun.get("early") if begun is not None else None
    if early is not None and is_training and EARLY_MLP3 and torch.is_grad_enabled():
        # The three anchor MLPs of the view, enqueued NOW on the buffer the level kernels (already in the queue) are writing:
        # their operands — the coded features in coding order, the visible anchors' rows of it, the camera — are all known here,
        # and the host has ~0.3 ms of node creation in front of it (hyper step, three level nodes, the mask's node) during which
        # the device would finish the level kernels and idle (profiles/r06_gpu_idle_gaps.txt).  The node is created later, by
        # _generate_unfused, around this launch (mlp.anchor_mlp3_rows(pre=)); if the step turns out different (the plan was
        # rebuilt, another expansion path), the launch is dropped and repeated there.
        mo_, mc_, mv_ = pc.get_opacity_mlp, pc.get_color_mlp, pc.get_cov_mlp
        if (mlp.anchor_mlp3_supported(mo_, mc_, mv_) and early["big"][0].shape[1] == 50
                and not anchor_gen.supported(mo_, mc_, mv_, pc.n_offsets, full_anchor)):
            from .context_model import _index_rows
            pos_e = _index_rows(early["cache"]["inv_perm"], vis_idx)
            begun["early_pos"] = (vis_idx, pos_e, early["cache"])      # (multi_scale_generating_visible hands the same rows on)
            early["mlp3"] = mlp.anchor_mlp3_rows_launch(early["big"][0], pos_e, anchor.detach(), viewpoint_camera.camera_center,
                                                        mo_, mc_, mv_)
    if use_context:
        # enqueued here, in front of the context model's count read-back: the host falls behind the device while it
        # waits there, and a gather the device still has queued covers part of the catch-up
        if late_mask is not None:
            from .context_model import _index_rows
            late_mask["vis_vals"] = _index_rows(late_mask["vals"], vis_idx)
            binary_grid_masks = None
        else:
            binary_grid_masks = sel(binary_all)
    if not use_context:
        # raw features of the visible anchors: left as (source, rows) when nothing is added to them, so that the
        # anchor-MLP kernel gathers them itself
        plain = not (is_training and 3000 < step <= 10000)
        feat = LazyRows(pc._anchor_feat, vis_idx) if (plain and pc._anchor_feat.is_cuda) else sel(pc._anchor_feat)
        grid_offsets = sel(pc._offset)
        grid_scaling = sel(pc.get_scaling)
        binary_grid_masks = sel(pc.get_mask)
        if is_training and 3000 < step <= 10000:                                        # :54-58
            if feat.is_cuda and grid_offsets.dim() == 3:
                # x + U(-1/2, 1/2) Q for the three tensors in ONE launch (the level kernel of the context model with a
                # zero step-size adjustment: Q = q0 (1 + tanh 0) = q0) instead of three uniform_ / mul / add chains over
                # [n,50], [n,6], [n,30]; the build's counter-based generator instead of torch's Philox stream (same
                # distribution), backward = identity
                from . import ctx_ops as _ctx
                n_vis = feat.shape[0]
                zero_adj = torch.zeros(n_vis, 3, dtype=torch.float32, device=feat.device)
                off_flat = grid_offsets.reshape(n_vis, grid_offsets.shape[1] * grid_offsets.shape[2])    # (explicit: n_vis may be 0)
                feat, grid_scaling, off2, _q = _ctx.noise_quant(feat, grid_scaling, off_flat, zero_adj,
                                                                (Q_FEAT, Q_SCALING, Q_OFFSETS))
                grid_offsets = off2.view(grid_offsets.shape)
            else:
                feat = feat + torch.empty_like(feat).uniform_(-0.5, 0.5) * Q_FEAT
                grid_scaling = grid_scaling + torch.empty_like(grid_scaling).uniform_(-0.5, 0.5) * Q_SCALING
                grid_offsets = grid_offsets + torch.empty_like(grid_offsets).uniform_(-0.5, 0.5) * Q_OFFSETS
    if is_training and step == 10000:                                                   # :60-61
        pc.update_anchor_bound()
    if use_context:                                                                     # :63-81 (train) / :83-101 (eval)
        # (get_mask_anchor, scene/gaussian_model.py:302-310, is "any offset alive" of the SAME mask values: both came
        # from one evaluation above instead of running the sigmoid / threshold / STE chain twice)
        res = multi_scale_generating_visible(pc, full_anchor, pc._hyper_latent, pc._anchor_feat, pc._offset,
                                             scaling_all if scaling_all is not None else pc.get_scaling, binary_all,
                                             mask_anchor_bool, vis_idx,
                                             training=is_training, predict_bpp=is_training, defer_feat=True,
                                             begun=begun, defer_rate=True)
        feat, grid_scaling, grid_offsets = res[:3]
        if is_training:
            rate_thunk = res[3]         # the rate model (:1657-1707) is enqueued behind the expansion's count (see below)
        if late_mask is not None:       # the level loop's nodes exist: now the mask's
            from .context_model import gather_unique_attach, gather_unique_pair_attach
            chosen = getattr(rate_thunk, "chosen_rows", None) if rate_thunk is not None else None
            if chosen is not None and MASK_PAIR_NODE and chosen.is_cuda:
                # the visible rows (for the expansion) and the rate subset's rows of the mask weights through ONE node: one
                # gradient buffer instead of two and the engine's add (context_model._GatherUniquePair)
                binary_grid_masks, m_chosen = gather_unique_pair_attach(binary_all(), vis_idx, late_mask["vis_vals"], chosen)
                rate_thunk = (lambda f, m: (lambda: f(m)))(rate_thunk, m_chosen)
            else:
                binary_grid_masks = gather_unique_attach(binary_all(), vis_idx, late_mask["vis_vals"])

    K = pc.n_offsets
    rate_out = []
    run_rate = (lambda: rate_out.extend(rate_thunk())) if rate_thunk is not None else None
    out = _generate_fused(viewpoint_camera, pc, anchor, feat, grid_scaling, grid_offsets, binary_grid_masks, K)
    if out is None:
        out = _generate_unfused(viewpoint_camera, pc, anchor, feat, grid_scaling, grid_offsets, binary_grid_masks, K,
                                between=run_rate, view=_view if is_training else None,
                                mlp_pre=early.get("mlp3") if early is not None else None)
    elif run_rate is not None:
        run_rate()
    if rate_out:
        bit_per_param, bit_per_feat_param, bit_per_scaling_param, bit_per_offsets_param, bpp_per_level = rate_out
    xyz, color, opacity, scaling, rot, neural_opacity, mask = out

    if is_training:                                                                      # :147-150
        return (xyz, color, opacity, scaling, rot, neural_opacity, mask, bit_per_param, 16, bit_per_feat_param,
                bit_per_scaling_param, bit_per_offsets_param, bpp_per_level)
    return xyz, color, opacity, scaling, rot, time_sub


class _ZeroPoints(torch.autograd.Function):
    """The reference's `screenspace_points` (gaussian_renderer/__init__.py:168: a zero [P,3] non-leaf tensor whose .grad
    receives the 2-D position gradients): a FRESH zero tensor per view, as in the reference — independent storage, so a caller
    that writes into one view's tensor does not touch another's (rounds 3-4 handed every view a slice of one pooled zero
    buffer, "read-only by contract") — wrapped in an autograd node so that it is a non-leaf tensor that requires
    grad without the reference's `+ 0` launch.  One fill of 12 B per Gaussian per view."""

    @staticmethod
    def forward(ctx, like, token):
        return torch.zeros(int(like.shape[0]), 3, dtype=torch.float32, device=like.device)

    @staticmethod
    def backward(ctx, _g):
        return None, None


_GRAD_TOKEN = {}


def _zero_points(xyz):
    tok = _GRAD_TOKEN.get(xyz.device)
    if tok is None:
        tok = _GRAD_TOKEN[xyz.device] = torch.zeros(1, device=xyz.device, requires_grad=True)
    return _ZeroPoints.apply(xyz.detach(), tok)


def _raster_settings(viewpoint_camera, pipe, bg_color, scaling_modifier):
    return GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=1, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=bool(getattr(pipe, "debug", False)))


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, visible_mask=None, retain_grad=False,
           step=0):                                                                      # :155-229
    """Render the scene.  Background tensor (bg_color) must be on the GPU."""
    is_training = pc.get_color_mlp.training
    view = None
    if is_training:
        # the expansion may run fused with the rasterizer's per-Gaussian stages (then `view.done` holds the rendered view
        # and xyz / color / opacity / rot are None: they never existed as tensors)
        view = ViewFusion(_raster_settings(viewpoint_camera, pipe, bg_color, scaling_modifier), retain_grad)
        (xyz, color, opacity, scaling, rot, neural_opacity, mask, bit_per_param, bit_per_anchor_param,
         bit_per_feat_param, bit_per_scaling_param, bit_per_offsets_param, bpp_per_level) = \
            generate_neural_gaussians(viewpoint_camera, pc, visible_mask, is_training=True, step=step, _view=view)
        if view.done is not None:
            rendered_image, radii, screenspace_points = view.done
            return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
                    "radii": radii, "selection_mask": mask, "neural_opacity": neural_opacity, "scaling": scaling,
                    "bit_per_param": bit_per_param, "bit_per_anchor_param": bit_per_anchor_param,
                    "bit_per_feat_param": bit_per_feat_param, "bit_per_scaling_param": bit_per_scaling_param,
                    "bit_per_offsets_param": bit_per_offsets_param, "bpp_per_level": bpp_per_level}
    else:
        xyz, color, opacity, scaling, rot, time_sub = generate_neural_gaussians(
            viewpoint_camera, pc, visible_mask, is_training=False, step=step)

    screenspace_points = _zero_points(xyz)       # :168 `torch.zeros_like(xyz, requires_grad=True) + 0`
    if retain_grad:
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass

    rasterizer = GaussianRasterizer(_raster_settings(viewpoint_camera, pipe, bg_color, scaling_modifier))
    rendered_image, radii = rasterizer(means3D=xyz, means2D=screenspace_points, shs=None, colors_precomp=color,
                                       opacities=opacity, scales=scaling, rotations=rot, cov3D_precomp=None)
    if is_training:
        return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
                "radii": radii, "selection_mask": mask, "neural_opacity": neural_opacity, "scaling": scaling,
                "bit_per_param": bit_per_param, "bit_per_anchor_param": bit_per_anchor_param,
                "bit_per_feat_param": bit_per_feat_param, "bit_per_scaling_param": bit_per_scaling_param,
                "bit_per_offsets_param": bit_per_offsets_param, "bpp_per_level": bpp_per_level}
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "time_sub": time_sub}


def prefilter_voxel(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):   # :232-287
    """Anchor-level frustum/size cull: bool[N]."""
    rs = _raster_settings(viewpoint_camera, pipe, bg_color, scaling_modifier)
    with torch.no_grad():
        means3D = pc.get_anchor
        # The reference evaluates get_scaling / get_rotation on all N rows and then reads three scale columns and
        # rotation row 0 (:262-266, :283: `rotations[[0], :].repeat(N, 1)`).  Both activations are row / element-wise,
        # so applying them to exactly what is read gives the same values: exp of 3 columns, normalisation of one row.
        # (rotation row 0 changes only when the optimizer touches _rotation: normalised once per version of the parameter)
        hit = getattr(pc, "_rot0_cache", None)
        if hit is not None and hit[0] is pc._rotation and hit[1] == pc._rotation._version:
            rot0 = hit[2]
        else:
            rot0 = pc.rotation_activation(pc._rotation[:1])
            pc._rot0_cache = (pc._rotation, pc._rotation._version, rot0)
        sc = pc._scaling
        if (means3D.is_cuda and sc.is_cuda and sc.dtype == torch.float32 and sc.dim() == 2 and sc.stride(1) == 1
                and means3D.dtype == torch.float32):
            # one launch (csrc/raster_geom.hip filter_voxel_kernel): exp of the three columns and the shared rotation are
            # applied inside, the bool mask comes out directly — no exp / repeat / zero-fill / compare launches
            from .rasterizer import _Cfg
            cfg = _Cfg(rs)
            N = int(means3D.shape[0])
            m = means3D.contiguous()
            r1 = rot0.reshape(-1).float().contiguous()
            vis = torch.empty(N, dtype=torch.bool, device=m.device)
            _lib.check(_lib.lib().cgs_filter_voxel(cfg.ref, N, _lib.ptr(m), sc.data_ptr(), int(sc.stride(0)),
                                                   0 if pc.decoded_version else 1, _lib.ptr(r1), _lib.ptr(vis),
                                                   _lib.current_stream()), "cgs_filter_voxel")
            return vis
        scales3 = sc[:, :3] if pc.decoded_version else torch.exp(sc[:, :3])
        rasterizer = GaussianRasterizer(rs)
        radii_pure = rasterizer.visible_filter(means3D=means3D, scales=scales3,
                                               rotations=rot0.repeat(means3D.shape[0], 1), cov3D_precomp=None)
    return radii_pure > 0
