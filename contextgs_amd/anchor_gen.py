"""generate_neural_gaussians' anchor -> Gaussian stage (gaussian_renderer/__init__.py:106-145) as ONE autograd node
over libcgs_hip.so's fused kernel family (csrc/anchor_gen.hip): the three anchor MLPs, the opacity mask, the survivor
compaction and the per-Gaussian tail.  Replaces `_AnchorMLP3Rows` + `_ExpandGaussians` (two nodes, five launches and
~2.9 KB of intermediates per anchor each way) when the heads have the reference's shape (K = 10 offsets).
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.nn as nn

from . import _lib
from .mlp import _ptr_array, _zeros_views, anchor_mlp3_supported

K_FUSED = 10
# Which calls take the fused node ("all" | "nograd" | "off"; every setting computes the same function).  Measured on
# MI355X, 1 M anchors (profiles/r03_anchor_gen_experiments.txt): the fused FORWARD matches the unfused pair with 60 % less
# HBM traffic, but the fused BACKWARD loses to the three-kernel pipeline (LDS float atomics run at ~0.6 lanes/clk), so
# training keeps the unfused pair and the fused node serves the no-grad calls (eval rendering, Test FPS).
MODE = os.environ.get("CGS_ANCHOR_GEN", "nograd")
ENABLED = MODE != "off"
FUSED_WGRAD = os.environ.get("CGS_ANCHOR_GEN_WGRAD", "0") != "0"


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def supported(mo: nn.Sequential, mc: nn.Sequential, mv: nn.Sequential, K: int, *tensors) -> bool:
    if MODE == "nograd" and torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in list(tensors) + [p for m in (mo, mc, mv) for p in m.parameters()]):
        return False
    return (ENABLED and K == K_FUSED and anchor_mlp3_supported(mo, mc, mv)
            and all(t is None or (t.is_cuda and t.dtype in (torch.float32, torch.int64)) for t in tensors))


class _AnchorGen(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat_src, feat_row, anchor_vis, cam, gs_src, off_src, geo_row, masks, *params):
        # params: (W1, b1, W2, b2) x (opacity, color, cov)
        L = _lib.lib()
        feat_src, anchor_vis, cam = _f32c(feat_src.detach()), _f32c(anchor_vis.detach()), _f32c(cam.detach()).reshape(-1)
        gs_src, masks = _f32c(gs_src.detach()), _f32c(masks.detach())
        off_src = _f32c(off_src.detach())
        off_shape = off_src.shape
        _lib.require_device(feat_src, anchor_vis, cam, gs_src, off_src, masks, feat_row, geo_row)
        n = int(anchor_vis.shape[0])
        assert feat_src.dim() == 2 and feat_src.shape[1] == 50 and cam.numel() == 3 and gs_src.shape[-1] == 6
        assert masks.numel() == n * K_FUSED and off_src.numel() == gs_src.shape[0] * 3 * K_FUSED
        for r, src in ((feat_row, feat_src), (geo_row, gs_src)):
            assert (r is None and src.shape[0] == n) or (r is not None and r.dtype == torch.int64 and r.numel() == n)
        p = [t.detach().contiguous() for t in params]
        W1, b1, W2, b2 = p[0::4], p[1::4], p[2::4], p[3::4]
        dev = feat_src.device
        need_grad = any(ctx.needs_input_grad)
        stream = _lib.current_stream()
        slots = n * K_FUSED
        neural_opacity = torch.empty(slots, 1, dtype=torch.float32, device=dev)
        mask_out = torch.empty(slots, dtype=torch.bool, device=dev)
        y_op = torch.empty(n, K_FUSED, dtype=torch.float32, device=dev) if need_grad else None
        bits = torch.empty(n, dtype=torch.int32, device=dev)
        base16 = torch.empty((n + 15) // 16 + 1, dtype=torch.int32, device=dev)
        cnt = C.c_int64(0)
        _lib.check(L.cgs_anchor_gen_count(_lib.ptr(feat_src), _lib.ptr(feat_row), _lib.ptr(anchor_vis), _lib.ptr(cam),
                                          _lib.ptr(masks), _lib.ptr(W1[0]), _lib.ptr(b1[0]), _lib.ptr(W2[0]), _lib.ptr(b2[0]),
                                          _lib.ptr(y_op), _lib.ptr(neural_opacity), _lib.ptr(mask_out), _lib.ptr(bits),
                                          _lib.ptr(base16), n, K_FUSED, C.byref(cnt), stream), "cgs_anchor_gen_count")
        P = int(cnt.value)
        xyz = torch.empty(P, 3, dtype=torch.float32, device=dev)
        color = torch.empty(P, 3, dtype=torch.float32, device=dev)
        opacity = torch.empty(P, 1, dtype=torch.float32, device=dev)
        scaling = torch.empty(P, 3, dtype=torch.float32, device=dev)
        rot = torch.empty(P, 4, dtype=torch.float32, device=dev)
        siginv = torch.empty(P, 4, dtype=torch.float32, device=dev) if need_grad else None
        _lib.check(L.cgs_anchor_gen_write(_lib.ptr(feat_src), _lib.ptr(feat_row), _lib.ptr(anchor_vis), _lib.ptr(cam),
                                          _lib.ptr(gs_src), _lib.ptr(off_src), _lib.ptr(geo_row), _lib.ptr(neural_opacity),
                                          _lib.ptr(bits), _lib.ptr(base16), _ptr_array(W1[1:]), _ptr_array(b1[1:]),
                                          _ptr_array(W2[1:]), _ptr_array(b2[1:]), _lib.ptr(xyz), _lib.ptr(color),
                                          _lib.ptr(opacity), _lib.ptr(scaling), _lib.ptr(rot), _lib.ptr(siginv), n, K_FUSED,
                                          stream), "cgs_anchor_gen_write")
        if need_grad:
            ctx.save_for_backward(feat_src, feat_row, anchor_vis, cam, gs_src, off_src, geo_row, masks, y_op, bits, base16,
                                  color, rot, siginv, *W1, *b1, *W2)
            ctx.off_shape = off_shape
            ctx.n = n
        ctx.mark_non_differentiable(mask_out)
        return xyz, color, opacity, scaling, rot, neural_opacity, mask_out

    @staticmethod
    def backward(ctx, g_xyz, g_color, g_opacity, g_scaling, g_rot, g_no, _g_mask):
        L = _lib.lib()
        sv = ctx.saved_tensors
        (feat_src, feat_row, anchor_vis, cam, gs_src, off_src, geo_row, masks, y_op, bits, base16, color, rot,
         siginv) = sv[:14]
        W1, b1, W2 = list(sv[14:17]), list(sv[17:20]), list(sv[20:23])
        n = ctx.n
        dev = feat_src.device
        P = int(color.shape[0])
        z = lambda t, w: torch.zeros(P, w, dtype=torch.float32, device=dev) if t is None else _f32c(t)
        g_xyz, g_color, g_opacity, g_scaling, g_rot = z(g_xyz, 3), z(g_color, 3), z(g_opacity, 1), z(g_scaling, 3), z(g_rot, 4)
        g_no = _f32c(g_no) if g_no is not None else None
        # rows of the sources no visible anchor reads keep a zero gradient; when every row is read: no fill
        alloc_f = torch.empty if n == feat_src.shape[0] else torch.zeros
        d_feat = alloc_f(feat_src.shape[0], 50, dtype=torch.float32, device=dev)
        alloc_g = torch.empty if n == gs_src.shape[0] else torch.zeros
        flat = alloc_g(gs_src.numel() + off_src.numel(), dtype=torch.float32, device=dev)
        d_gs, d_off = flat[:gs_src.numel()].view_as(gs_src), flat[gs_src.numel():].view(ctx.off_shape)
        d_anchor = torch.empty(n, 3, dtype=torch.float32, device=dev)
        d_mask = torch.empty_like(masks)
        views = _zeros_views(dev, (150, 54), (150,), *[tuple(w.shape) for w in W2], *[(w.shape[0],) for w in W2])
        dW1cat, db1cat, dW2, db2 = views[0], views[1], list(views[2:5]), list(views[5:8])
        fused = 1 if FUSED_WGRAD else 0
        ws = torch.empty(int(L.cgs_anchor_gen_bwd_scratch_bytes(n, fused)), dtype=torch.uint8, device=dev)
        _lib.check(L.cgs_anchor_gen_backward(
            _lib.ptr(feat_src), _lib.ptr(feat_row), _lib.ptr(anchor_vis), _lib.ptr(cam), _lib.ptr(gs_src), _lib.ptr(off_src),
            _lib.ptr(geo_row), _lib.ptr(masks), _lib.ptr(y_op), _lib.ptr(bits), _lib.ptr(base16), _lib.ptr(color),
            _lib.ptr(rot), _lib.ptr(siginv), _lib.ptr(g_xyz), _lib.ptr(g_color), _lib.ptr(g_opacity), _lib.ptr(g_scaling),
            _lib.ptr(g_rot), _lib.ptr(g_no), _ptr_array(W1), _ptr_array(b1), _ptr_array(W2), _lib.ptr(d_feat),
            _lib.ptr(d_anchor), _lib.ptr(d_gs), _lib.ptr(d_off), _lib.ptr(d_mask), _lib.ptr(dW1cat), _lib.ptr(db1cat),
            _ptr_array(dW2), _ptr_array(db2), n, K_FUSED, fused, _lib.ptr(ws), ws.numel(), _lib.current_stream()),
            "cgs_anchor_gen_backward")
        need = ctx.needs_input_grad
        grads = [d_feat if need[0] else None, None, d_anchor if need[2] else None, None, d_gs if need[4] else None,
                 d_off if need[5] else None, None, d_mask if need[7] else None]
        for i in range(3):
            grads += [dW1cat[50 * i:50 * (i + 1)], db1cat[50 * i:50 * (i + 1)], dW2[i], db2[i]]
        return tuple(grads)


def anchor_gen(feat_src, feat_row, anchor_vis, cam_center, gs_src, off_src, geo_row, masks,
               mo: nn.Sequential, mc: nn.Sequential, mv: nn.Sequential):
    """(xyz, color, opacity, scaling, rot, neural_opacity, selection_mask) of the visible anchors: anchor r has the
    MLP input [feat_src[feat_row[r]] | unit view vector | distance] and the geometry rows gs_src / off_src[geo_row[r]]
    (a row index of None = identity), masks [n, K] its offset mask."""
    params = []
    for s in (mo, mc, mv):
        params += [s[0].weight, s[0].bias, s[2].weight, s[2].bias]
    return _AnchorGen.apply(feat_src, feat_row, anchor_vis, cam_center, gs_src, off_src, geo_row, masks, *params)
