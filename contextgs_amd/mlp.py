"""`mlp2(x, seq)`: the nn.Sequential(Linear, ReLU, Linear [, Tanh | Sigmoid]) modules of the
hot path (scene/gaussian_model.py:153-188) evaluated by libcgs_hip.so's fused fp32-MFMA
kernels (csrc/mlp.hip) — forward, input gradient and weight/bias gradients — instead of
4-6 rocBLAS/elementwise launches each way.  Falls back to nothing: shapes without a kernel
instance raise.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _lib

_ACT = {None: 0, nn.Tanh: 1, nn.Sigmoid: 2}
_RECOMPUTE = {(71, 100, 3, 0), (15, 100, 3, 0)}
RECOMPUTE_HIDDEN = os.environ.get("CGS_MLP_RECOMPUTE", "1") != "0"      # tuning / A-B knob


def _zeros_views(dev, *shapes):
    """Zero-initialised tensors of the given shapes carved from ONE buffer (one fill launch)."""
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    return [t.view(s) for t, s in zip(torch.split(flat, sizes), shapes)]


def _wgrad_workspace(dev) -> torch.Tensor:
    """Device scratch for the two-pass (atomics-free, deterministic) weight-gradient reduction; taken from the
    caching allocator per call so that it is stream-safe."""
    return torch.empty(int(_lib.lib().cgs_mlp_wgrad_scratch_bytes()), dtype=torch.uint8, device=dev)


# ---- weight gradients deferred to the end of the backward (data-parallel steps) ------------------------------------
class _Deferred:
    """Queue of weight-gradient launches that the backward nodes below leave for the END of the backward pass.

    On one GPU a node launches its weight-gradient products right behind its data-gradient kernel.  In a data-parallel
    step the per-anchor gradients (0.44 GB at 1 M anchors) become final in the last nodes of the graph, so their
    all-reduce has nothing left to hide behind — unless the ≈ 1.1 ms of weight-gradient kernels, which nothing in
    the backward depends on, run AFTER it has been issued.  With `defer_weight_gradients(True)` the nodes call the
    data-only form of their C entry point, keep the operands alive in a closure, and an autograd-engine callback
    (end of `backward()`) first runs the `before_flush` hooks (`dist.GradientSync` issues its remaining per-anchor
    collectives there), then launches the queued products on the compute stream and accumulates the results into
    the parameters' `.grad` exactly as autograd would have (same kernels, same order: bit-identical gradients)."""
    on = os.environ.get("CGS_DEFER_WGRAD", "0") == "1"
    queue: list = []
    armed = False
    before_flush: list = []
    staged: dict = {}


def defer_weight_gradients(on: bool = True) -> bool:
    """Switch the deferral on / off; returns the previous setting."""
    prev, _Deferred.on = _Deferred.on, bool(on)
    return prev


def add_before_flush_hook(fn):
    _Deferred.before_flush.append(fn)
    return fn


def remove_before_flush_hook(fn):
    if fn in _Deferred.before_flush:
        _Deferred.before_flush.remove(fn)


def _can_defer(params) -> bool:
    # only leaf parameters: a gradient for anything else has to travel through autograd
    return _Deferred.on and all(isinstance(p, torch.Tensor) and p.is_leaf and p.requires_grad for p in params)


def _flush_deferred():
    _Deferred.armed = False
    jobs, _Deferred.queue = _Deferred.queue, []
    for hook in list(_Deferred.before_flush):
        hook()
    for job in jobs:
        job()
    # like the engine's input buffer of an AccumulateGrad node: the contributions of THIS backward are summed first (in
    # node order), then added to an existing .grad
    staged, _Deferred.staged = _Deferred.staged, {}
    with torch.no_grad():
        for p, g in staged.values():
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)


def _drop_stale_deferred():
    """Called by the MLP nodes' FORWARD: a queue that is still armed then belongs to a backward that died before the engine
    ran its callbacks (an exception in some node) — its jobs hold that step's buffers and must neither run nor keep the next
    backward from arming a callback of its own."""
    if _Deferred.armed or _Deferred.queue or _Deferred.staged:
        _Deferred.armed = False
        _Deferred.queue = []
        _Deferred.staged = {}


def _defer(job):
    _Deferred.queue.append(job)
    if not _Deferred.armed:
        _Deferred.armed = True
        torch.autograd.Variable._execution_engine.queue_callback(_flush_deferred)


def _accumulate(params, grads):
    with torch.no_grad():
        for p, g in zip(params, grads):
            have = _Deferred.staged.get(id(p))
            _Deferred.staged[id(p)] = (p, g if have is None else have[1].add_(g))


_SUPPORTED = {(54, 50, 10, 1), (54, 50, 30, 2), (54, 50, 70, 0), (71, 100, 175, 0), (15, 100, 175, 0), (71, 100, 3, 0),
              (15, 100, 3, 0)}


def _describe(seq: nn.Sequential):
    l1, l2 = seq[0], seq[2]
    act = _ACT[type(seq[3])] if len(seq) > 3 else 0
    return l1, l2, act


def supported(seq: nn.Sequential) -> bool:
    try:
        l1, l2, act = _describe(seq)
    except Exception:
        return False
    return (l1.in_features, l1.out_features, l2.out_features, act) in _SUPPORTED


class _MLP2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, act):
        L = _lib.lib()
        _drop_stale_deferred()
        _lib.require_device(x, W1, W2)
        x = x.contiguous() if x.dtype == torch.float32 else x.float().contiguous()
        W1c, b1c, W2c, b2c = (t.detach().contiguous() for t in (W1, b1, W2, b2))
        n, in_f = x.shape
        hid, out = W1c.shape[0], W2c.shape[0]
        need_grad = any(ctx.needs_input_grad[:5])
        y = torch.empty(n, out, dtype=torch.float32, device=x.device)
        # tiny-output shapes: the hidden layer is recomputed by the backward instead of stored (csrc/mlp_small.hip)
        recompute = RECOMPUTE_HIDDEN and (in_f, hid, out, act) in _RECOMPUTE
        h = torch.empty(n, hid, dtype=torch.float32, device=x.device) if (need_grad and not recompute) else None
        _lib.check(L.cgs_mlp2_forward(in_f, hid, out, act, _lib.ptr(x), in_f, _lib.ptr(W1c), _lib.ptr(b1c), _lib.ptr(W2c),
                                      _lib.ptr(b2c), _lib.ptr(y), out, _lib.ptr(h), n, _lib.current_stream()),
                   "cgs_mlp2_forward")
        if need_grad:
            ctx.save_for_backward(x, W1c, W2c, y, h if h is not None else b1c)
        ctx.act, ctx.recompute = act, recompute
        ctx.params = (W1, b1, W2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, W1, W2, y, h = ctx.saved_tensors
        act = ctx.act
        b1 = None
        if ctx.recompute:
            b1, h = h, None
        n, in_f = x.shape
        hid, out = W1.shape[0], W2.shape[0]
        dev = x.device
        dy = dy.contiguous() if dy.dtype == torch.float32 else dy.float().contiguous()
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty(n, in_f, dtype=torch.float32, device=dev) if need_dx else None
        dz1 = torch.empty(n, hid, dtype=torch.float32, device=dev)
        dz2 = torch.empty(n, out, dtype=torch.float32, device=dev) if act != 0 else None
        dW1, db1, dW2, db2 = _zeros_views(dev, (hid, in_f), (hid,), (out, hid), (out,))   # one fill for the four
        ws = _wgrad_workspace(dev)
        defer = _can_defer(ctx.params) and n > 0
        inside = defer and h is None          # the recomputing kernel accumulates the second layer itself
        wg = lambda t, keep: _lib.ptr(t) if (not defer or keep) else None
        _lib.check(L.cgs_mlp2_backward(in_f, hid, out, act, _lib.ptr(x), in_f, _lib.ptr(W1), _lib.ptr(b1), _lib.ptr(W2), _lib.ptr(y),
                                       _lib.ptr(dy), out, _lib.ptr(h), _lib.ptr(dx), in_f, 0, _lib.ptr(dz1), _lib.ptr(dz2),
                                       wg(dW1, False), wg(db1, False), wg(dW2, inside), wg(db2, inside), n,
                                       _lib.ptr(ws), ws.numel(), _lib.current_stream()), "cgs_mlp2_backward")
        if not defer:
            return dx, dW1, db1, dW2, db2, None
        p2 = dz2 if dz2 is not None else dy
        params = ctx.params

        def job():
            _lib.check(L.cgs_mlp2_wgrad(in_f, hid, out, _lib.ptr(x), in_f, _lib.ptr(h), _lib.ptr(p2), out, _lib.ptr(dz1),
                                        _lib.ptr(dW1), _lib.ptr(db1), _lib.ptr(dW2), _lib.ptr(db2), n, _lib.ptr(ws), ws.numel(),
                                        _lib.current_stream()), "cgs_mlp2_wgrad")
            _accumulate(params, (dW1, db1, dW2, db2))
        _defer(job)
        return dx, None, None, None, None, None


class _LevelMLP(torch.autograd.Function):
    """mlp_grid[level] (scene/gaussian_model.py:1600) with BOTH of its training consumers in one autograd node:
    the last `out - n_stat` outputs (the quantisation-step adjustments) on every row of x, all outputs on the rows
    `loc` (the rate subset, :1658-1669).  One node instead of two MLP nodes over weight slices: the backward fills
    the weight gradients of the whole module once (no slice zero-fill / copy / add per consumer) and merges the
    subset's input gradient into the full one row-wise."""

    @staticmethod
    def forward(ctx, x, loc, W1, b1, W2, b2, n_stat):
        from . import ctx_ops
        L = _lib.lib()
        _drop_stale_deferred()
        _lib.require_device(x, W1, W2, loc)
        x = x.contiguous() if x.dtype == torch.float32 else x.float().contiguous()
        W1c, b1c, W2c, b2c = (t.detach().contiguous() for t in (W1, b1, W2, b2))
        n, in_f = x.shape
        hid, out = W1c.shape[0], W2c.shape[0]
        nq, m = out - n_stat, int(loc.shape[0])
        need_grad = any(ctx.needs_input_grad)
        stream = _lib.current_stream()
        qadj = torch.empty(n, nq, dtype=torch.float32, device=x.device)
        # rows n_stat.. of the second layer are a contiguous block of W2 / b2: plain pointer offsets, no slice op
        _lib.check(L.cgs_mlp2_forward(in_f, hid, nq, 0, _lib.ptr(x), in_f, _lib.ptr(W1c), _lib.ptr(b1c),
                                      W2c.data_ptr() + 4 * n_stat * hid, b2c.data_ptr() + 4 * n_stat, _lib.ptr(qadj), nq,
                                      None, n, stream), "cgs_mlp2_forward")
        x_sub = ctx_ops.gather_rows_nograd(x, loc)
        pred = torch.empty(m, out, dtype=torch.float32, device=x.device)
        h_sub = torch.empty(m, hid, dtype=torch.float32, device=x.device) if need_grad else None
        _lib.check(L.cgs_mlp2_forward(in_f, hid, out, 0, _lib.ptr(x_sub), in_f, _lib.ptr(W1c), _lib.ptr(b1c), _lib.ptr(W2c),
                                      _lib.ptr(b2c), _lib.ptr(pred), out, _lib.ptr(h_sub), m, stream), "cgs_mlp2_forward")
        if need_grad:
            ctx.save_for_backward(x, x_sub, loc, W1c, b1c, W2c, h_sub)
        ctx.n_stat = n_stat
        ctx.params = (W1, b1, W2, b2)
        return qadj, pred

    @staticmethod
    def backward(ctx, d_q, d_pred):
        L = _lib.lib()
        x, x_sub, loc, W1, b1, W2, h_sub = ctx.saved_tensors
        n_stat = ctx.n_stat
        n, in_f = x.shape
        hid, out = W1.shape[0], W2.shape[0]
        nq, m = out - n_stat, int(loc.shape[0])
        dev = x.device
        f32c = lambda t: t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()
        need_dx = ctx.needs_input_grad[0]
        stream = _lib.current_stream()
        dW1, db1, dW2, db2 = _zeros_views(dev, (hid, in_f), (hid,), (out, hid), (out,))   # one fill for the module
        ws = _wgrad_workspace(dev)
        dx = None
        defer = _can_defer(ctx.params)
        wg = lambda t: None if defer else _lib.ptr(t)
        jobs = []
        if d_q is not None:
            dx = torch.empty(n, in_f, dtype=torch.float32, device=dev) if need_dx else None
            dz1 = torch.empty(n, hid, dtype=torch.float32, device=dev)
            d_q = f32c(d_q)
            # (recomputing kernel: the step-size rows of the second layer are accumulated inside it, deferred or not)
            _lib.check(L.cgs_mlp2_backward(in_f, hid, nq, 0, _lib.ptr(x), in_f, _lib.ptr(W1), _lib.ptr(b1),
                                           W2.data_ptr() + 4 * n_stat * hid, None, _lib.ptr(d_q), nq, None, _lib.ptr(dx), in_f, 0,
                                           _lib.ptr(dz1), None, wg(dW1), wg(db1), dW2.data_ptr() + 4 * n_stat * hid,
                                           db2.data_ptr() + 4 * n_stat, n, _lib.ptr(ws), ws.numel(), stream), "cgs_mlp2_backward")
            if defer and n > 0:
                jobs.append(lambda: _lib.check(L.cgs_mlp2_wgrad(
                    in_f, hid, nq, _lib.ptr(x), in_f, None, None, 0, _lib.ptr(dz1), _lib.ptr(dW1), _lib.ptr(db1), None, None, n,
                    _lib.ptr(ws), ws.numel(), _lib.current_stream()), "cgs_mlp2_wgrad"))
        if d_pred is not None and m > 0:
            d_pred = f32c(d_pred)
            dz1s = torch.empty(m, hid, dtype=torch.float32, device=dev)
            if need_dx and dx is None:
                dx = torch.zeros(n, in_f, dtype=torch.float32, device=dev)
            # the subset's input gradient is ADDED into rows loc (unique) of dx by the kernel's own store
            # (cgs_mlp2_backward_rows): no [m, in] temporary, no index_add_ launch
            _lib.check(L.cgs_mlp2_backward_rows(in_f, hid, out, 0, _lib.ptr(x_sub), in_f, _lib.ptr(W1), None, _lib.ptr(W2), None,
                                                _lib.ptr(d_pred), out, _lib.ptr(h_sub), _lib.ptr(dx) if need_dx else None, in_f, 1,
                                                _lib.ptr(loc), _lib.ptr(dz1s), None, wg(dW1), wg(db1), wg(dW2),
                                                wg(db2), m, _lib.ptr(ws), ws.numel(), stream), "cgs_mlp2_backward_rows")
            if defer:
                jobs.append(lambda: _lib.check(L.cgs_mlp2_wgrad(
                    in_f, hid, out, _lib.ptr(x_sub), in_f, _lib.ptr(h_sub), _lib.ptr(d_pred), out, _lib.ptr(dz1s), _lib.ptr(dW1),
                    _lib.ptr(db1), _lib.ptr(dW2), _lib.ptr(db2), m, _lib.ptr(ws), ws.numel(), _lib.current_stream()),
                    "cgs_mlp2_wgrad"))
        if not defer:
            return dx, None, dW1, db1, dW2, db2, None
        params = ctx.params

        def job():
            for j in jobs:
                j()
            _accumulate(params, (dW1, db1, dW2, db2))
        _defer(job)
        return dx, None, None, None, None, None, None


def level_mlp(x: torch.Tensor, loc: torch.Tensor, seq: nn.Sequential, n_stat: int):
    """(seq(x)[:, n_stat:], seq(x[loc])) for a supported Sequential(Linear, ReLU, Linear), one autograd node."""
    l1, l2, act = _describe(seq)
    in_f, hid, out = l1.in_features, l1.out_features, l2.out_features
    if act != 0 or (in_f, hid, out, 0) not in _SUPPORTED or (in_f, hid, out - n_stat, 0) not in _SUPPORTED:
        raise NotImplementedError(f"no fused MLP kernels for {(in_f, hid, out)} split at {n_stat}")
    return _LevelMLP.apply(x, loc, l1.weight, l1.bias, l2.weight, l2.bias, int(n_stat))


def mlp2_weights(x: torch.Tensor, W1, b1, W2, b2, act: int = 0) -> torch.Tensor:
    """Same kernels on explicit weight tensors (e.g. a row slice of the second layer)."""
    key = (W1.shape[1], W1.shape[0], W2.shape[0], act)
    if key not in _SUPPORTED:
        raise NotImplementedError(f"no fused MLP kernel for {key}; instantiate it in csrc/mlp.hip")
    return _MLP2.apply(x, W1, b1, W2, b2, act)


def mlp2(x: torch.Tensor, seq: nn.Sequential) -> torch.Tensor:
    """seq(x) for a supported Sequential(Linear, ReLU, Linear[, act])."""
    l1, l2, act = _describe(seq)
    key = (l1.in_features, l1.out_features, l2.out_features, act)
    if key not in _SUPPORTED:
        raise NotImplementedError(f"no fused MLP kernel for {key}; instantiate it in csrc/mlp.hip")
    return _MLP2.apply(x, l1.weight, l1.bias, l2.weight, l2.bias, act)


# ---- the three anchor MLPs in one launch (csrc/mlp3.hip) ---------------------------------------------
import ctypes as _C


_M3_LAYOUT = None


def _m3_layout():
    """(row stride of Hcat, row stride of X_out, row stride of dZ1cat, column pitch of a head in dZ1cat / dW1cat / db1cat)
    from the library (cgs_anchor_mlp3_layout)."""
    global _M3_LAYOUT
    if _M3_LAYOUT is None:
        out = (_C.c_int * 4)()
        _lib.check(_lib.lib().cgs_anchor_mlp3_layout(out), "cgs_anchor_mlp3_layout")
        _M3_LAYOUT = tuple(int(v) for v in out)
    return _M3_LAYOUT


def _ptr_array(tensors):
    return (_C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def anchor_mlp3_supported(mo: nn.Sequential, mc: nn.Sequential, mv: nn.Sequential) -> bool:
    try:
        keys = [(s[0].in_features, s[0].out_features, s[2].out_features, _describe(s)[2]) for s in (mo, mc, mv)]
    except Exception:
        return False
    return keys == [(54, 50, 10, 1), (54, 50, 30, 2), (54, 50, 70, 0)]


def _m3_wgrad_job(x, ldx, hcat, dz1, dz2_op, dz2_color, g_cov, dW1cat, db1cat, dW2, db2, n, ws, params, wgrads):
    """The deferred weight-gradient launch of an anchor-MLP backward node (operands kept alive by the closure)."""
    def job():
        _lib.check(_lib.lib().cgs_anchor_mlp3_wgrad(
            _lib.ptr(x), ldx, _lib.ptr(hcat), _lib.ptr(dz1), _lib.ptr(dz2_op), _lib.ptr(dz2_color), _lib.ptr(g_cov),
            _lib.ptr(dW1cat), _lib.ptr(db1cat), _ptr_array(dW2), _ptr_array(db2), n, _lib.ptr(ws), ws.numel(),
            _lib.current_stream()), "cgs_anchor_mlp3_wgrad")
        _accumulate(params, wgrads)
    return job


class _AnchorMLP3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, *params):
        # params: (W1, b1, W2, b2) x (opacity, color, cov)
        L = _lib.lib()
        _drop_stale_deferred()
        _lib.require_device(x)
        x = x.contiguous() if x.dtype == torch.float32 else x.float().contiguous()
        p = [t.detach().contiguous() for t in params]
        W1, b1, W2, b2 = p[0::4], p[1::4], p[2::4], p[3::4]
        n = x.shape[0]
        dev = x.device
        need_grad = any(ctx.needs_input_grad)
        y_op = torch.empty(n, 10, dtype=torch.float32, device=dev)
        y_color = torch.empty(n, 30, dtype=torch.float32, device=dev)
        y_cov = torch.empty(n, 70, dtype=torch.float32, device=dev)
        hcat = torch.empty(n, _m3_layout()[0], dtype=torch.float32, device=dev) if need_grad else None
        _lib.check(L.cgs_anchor_mlp3_forward(_lib.ptr(x), x.shape[1], _ptr_array(W1), _ptr_array(b1), _ptr_array(W2),
                                             _ptr_array(b2), _lib.ptr(y_op), _lib.ptr(y_color), _lib.ptr(y_cov),
                                             _lib.ptr(hcat), n, _lib.current_stream()), "cgs_anchor_mlp3_forward")
        if need_grad:
            ctx.save_for_backward(x, y_op, y_color, hcat, *W1, *W2)
        ctx.params = params
        return y_op, y_color, y_cov

    @staticmethod
    def backward(ctx, g_op, g_color, g_cov):
        L = _lib.lib()
        saved = ctx.saved_tensors
        x, y_op, y_color, hcat = saved[:4]
        W1, W2 = list(saved[4:7]), list(saved[7:10])
        n = x.shape[0]
        dev = x.device
        z = lambda t, shape: torch.zeros(shape, dtype=torch.float32, device=dev) if t is None else (
            t.contiguous() if t.dtype == torch.float32 else t.float().contiguous())
        g_op, g_color, g_cov = z(g_op, (n, 10)), z(g_color, (n, 30)), z(g_cov, (n, 70))
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty(n, x.shape[1], dtype=torch.float32, device=dev) if need_dx else None
        _hld, _xld, gld, hp = _m3_layout()
        dz1 = torch.empty(n, gld, dtype=torch.float32, device=dev)
        dz2_op = torch.empty(n, 10, dtype=torch.float32, device=dev)
        dz2_color = torch.empty(n, 30, dtype=torch.float32, device=dev)
        views = _zeros_views(dev, (gld, 54), (gld,), *[tuple(w.shape) for w in W2], *[(w.shape[0],) for w in W2])
        dW1cat, db1cat, dW2, db2 = views[0], views[1], list(views[2:5]), list(views[5:8])
        ws = _wgrad_workspace(dev)
        defer = _can_defer(ctx.params) and n > 0
        _lib.check(L.cgs_anchor_mlp3_backward(
            _lib.ptr(x), x.shape[1], _ptr_array(W1), _ptr_array(W2), _lib.ptr(y_op), _lib.ptr(y_color), _lib.ptr(g_op),
            _lib.ptr(g_color), _lib.ptr(g_cov), _lib.ptr(hcat), _lib.ptr(dx), x.shape[1], _lib.ptr(dz1), _lib.ptr(dz2_op),
            _lib.ptr(dz2_color), None if defer else _lib.ptr(dW1cat), None if defer else _lib.ptr(db1cat),
            None if defer else _ptr_array(dW2), None if defer else _ptr_array(db2), n,
            _lib.ptr(ws), ws.numel(), _lib.current_stream()), "cgs_anchor_mlp3_backward")
        wgrads = []
        for i in range(3):
            wgrads += [dW1cat[hp * i:hp * i + 50], db1cat[hp * i:hp * i + 50], dW2[i], db2[i]]
        if not defer:
            return (dx, *wgrads)
        _defer(_m3_wgrad_job(x, x.shape[1], hcat, dz1, dz2_op, dz2_color, g_cov, dW1cat, db1cat, dW2, db2, n, ws, ctx.params,
                             wgrads))
        return (dx,) + (None,) * 12


# A/B knob, default: the forward KEEPS its assembled rows.  CGS_M3_KEEP_X=0: it does not (forward 480 -> 438 us at 1 M anchors:
# it is bound by the bytes it stores) and the fused backward assembles them again (903 -> 957 us: one wave per SIMD, every
# instruction of the re-gather is added time) — a net loss of ~12 us per step, same box, two pairs: profiles/r06_mlp3_xout_ab.txt
KEEP_X_OFF = os.environ.get("CGS_M3_KEEP_X", "1") == "0"
# (round 6) the forward's private hand-over buffer Hcat in fragment-major form: every store of the forward and load of the fused
# backward one contiguous KB (include/cgs.h, cgs_anchor_mlp3_forward_rows_t): forward 456 -> 421 us at 1 M anchors.  Off when the
# weight gradients are deferred (the separate weight-gradient launch reads row-major buffers).  CGS_M3_TILED=0: row-major (A/B knob).
M3_TILED = os.environ.get("CGS_M3_TILED", "1") != "0"
_M3_REGATHER_MAX_ROWS = 4_000_000                                # the fused backward's row limit (32-bit buffer offsets)


class _AnchorMLP3Rows(torch.autograd.Function):
    """The three anchor MLPs on rows assembled inside the kernel: [feat_src[src_row] | view direction | distance]
    (gaussian_renderer/__init__.py:106-127); backward scatters the feature gradient into the source rows and pulls the
    view gradient back to the anchors — no [n,54] gather / scatter kernels around the MLP."""

    @staticmethod
    def launch(feat_src, src_row, anchor_vis, cam, params, need_grad):
        """The forward's launch on plain values: a dict forward() accepts as `pre`.  The renderer calls it EARLY — right after the
        visible-anchor list is known, on the buffer the step's level kernels are writing — so that the 0.5 ms kernel is in the
        queue while the host creates the context model's autograd nodes (renderer.generate_neural_gaussians)."""
        L = _lib.lib()
        _drop_stale_deferred()
        f32c = lambda t: t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()
        feat_src, anchor_vis, cam = f32c(feat_src.detach()), f32c(anchor_vis.detach()), f32c(cam.detach()).reshape(-1)
        _lib.require_device(feat_src, anchor_vis, cam, src_row)
        assert feat_src.dim() == 2 and feat_src.shape[1] == 50 and cam.numel() == 3 and src_row.dtype == torch.int64
        p = [t.detach().contiguous() for t in params]
        W1, b1, W2, b2 = p[0::4], p[1::4], p[2::4], p[3::4]
        n = int(src_row.shape[0])
        dev = feat_src.device
        y_op = torch.empty(n, 10, dtype=torch.float32, device=dev)
        y_color = torch.empty(n, 30, dtype=torch.float32, device=dev)
        y_cov = torch.empty(n, 70, dtype=torch.float32, device=dev)
        hld, xld, _gld, _gp = _m3_layout()
        tiled = bool(M3_TILED and need_grad and not _Deferred.on and 0 < n <= _M3_REGATHER_MAX_ROWS)
        n_buf = (n + 15) // 16 * 16 if tiled else n           # (whole 16-row tiles)
        hcat = torch.empty(n_buf, hld, dtype=torch.float32, device=dev) if need_grad else None
        # (round 6) the assembled input rows are kept only for the deferred weight-gradient launch of the multi-GPU step; the
        # fused backward of the single-GPU step assembles them again (216 of the 1256 bytes per anchor the forward is bound by)
        keep_x = need_grad and (_Deferred.on or n > _M3_REGATHER_MAX_ROWS or not KEEP_X_OFF)
        x = torch.empty(n, xld, dtype=torch.float32, device=dev) if keep_x else None
        _lib.check(L.cgs_anchor_mlp3_forward_rows_t(_lib.ptr(feat_src), _lib.ptr(src_row), _lib.ptr(anchor_vis), _lib.ptr(cam),
                                                    _lib.ptr(x), _ptr_array(W1), _ptr_array(b1), _ptr_array(W2), _ptr_array(b2),
                                                    _lib.ptr(y_op), _lib.ptr(y_color), _lib.ptr(y_cov), _lib.ptr(hcat), n, int(tiled),
                                                    _lib.current_stream()), "cgs_anchor_mlp3_forward_rows")
        return dict(feat_src=feat_src, src_row=src_row, anchor_vis=anchor_vis, cam=cam, W1=W1, W2=W2, y=(y_op, y_color, y_cov), hcat=hcat,
                    x=x, keep_x=keep_x, need_grad=need_grad, deferred=_Deferred.on, tiled=tiled,
                    key=(feat_src.data_ptr(), tuple(feat_src.shape), anchor_vis.data_ptr(), n))

    @staticmethod
    def forward(ctx, feat_src, src_row, anchor_vis, cam, pre, *params):
        ctx.set_materialize_grads(False)
        need_grad = any(ctx.needs_input_grad)
        if pre is not None:
            # an early launch is taken only if it ran on exactly these operands (else: dropped, launched again)
            fs_, av_ = feat_src.detach(), anchor_vis.detach()
            if not (pre["src_row"] is src_row and pre["need_grad"] == need_grad and pre["deferred"] == _Deferred.on
                    and fs_.dtype == torch.float32 and fs_.is_contiguous() and av_.dtype == torch.float32 and av_.is_contiguous()
                    and pre["key"] == (fs_.data_ptr(), tuple(fs_.shape), av_.data_ptr(), int(src_row.shape[0]))):
                pre = None
        if pre is None:
            pre = _AnchorMLP3Rows.launch(feat_src, src_row, anchor_vis, cam, params, need_grad)
        feat_src, anchor_vis, cam = pre["feat_src"], pre["anchor_vis"], pre["cam"]
        W1, W2 = pre["W1"], pre["W2"]
        y_op, y_color, y_cov = pre["y"]
        hcat, x, keep_x = pre["hcat"], pre["x"], pre["keep_x"]
        if need_grad:
            ctx.save_for_backward(x if keep_x else feat_src, src_row, anchor_vis, cam, y_op, y_color, hcat, *W1, *W2)
            ctx.n_src = int(feat_src.shape[0])
            ctx.keep_x = keep_x
            ctx.tiled = bool(pre.get("tiled", False))
        ctx.params = params
        return y_op, y_color, y_cov

    @staticmethod
    def backward(ctx, g_op, g_color, g_cov):
        L = _lib.lib()
        saved = ctx.saved_tensors
        x, src_row, anchor_vis, cam, y_op, y_color, hcat = saved[:7]
        feat_src = None
        if not ctx.keep_x:
            feat_src, x = x, None
        W1, W2 = list(saved[7:10]), list(saved[10:13])
        n = int(src_row.shape[0])
        dev = hcat.device
        z = lambda t, shape: torch.zeros(shape, dtype=torch.float32, device=dev) if t is None else (
            t.contiguous() if t.dtype == torch.float32 else t.float().contiguous())
        g_op, g_color, g_cov = z(g_op, (n, 10)), z(g_color, (n, 30)), z(g_cov, (n, 70))
        # rows of the source no visible anchor reads keep a zero gradient (ctx_ops.zero_unlisted_rows writes only those rows —
        # a 200 MB fill at 1 M anchors otherwise); when every row is read there is nothing to zero
        d_src = torch.empty(ctx.n_src, 50, dtype=torch.float32, device=dev)
        if n != ctx.n_src:
            from .ctx_ops import zero_unlisted_rows
            zero_unlisted_rows(src_row, ctx.n_src, [d_src])
        d_anchor = torch.empty(n, 3, dtype=torch.float32, device=dev)
        _hld, _xld, gld, hp = _m3_layout()
        dz1 = torch.empty(n, gld, dtype=torch.float32, device=dev)
        dz2_op = torch.empty(n, 10, dtype=torch.float32, device=dev)
        dz2_color = torch.empty(n, 30, dtype=torch.float32, device=dev)
        views = _zeros_views(dev, (gld, 54), (gld,), *[tuple(w.shape) for w in W2], *[(w.shape[0],) for w in W2])
        dW1cat, db1cat, dW2, db2 = views[0], views[1], list(views[2:5]), list(views[5:8])
        ws = _wgrad_workspace(dev)
        # (the tiled hand-over is read by the fused data + weight-gradient kernel only: never deferred)
        defer = _can_defer(ctx.params) and n > 0 and x is not None and not ctx.tiled
        _lib.check(L.cgs_anchor_mlp3_backward_rows_t(
            _lib.ptr(x), _lib.ptr(feat_src), _lib.ptr(src_row), _lib.ptr(anchor_vis), _lib.ptr(cam), _ptr_array(W1), _ptr_array(W2), _lib.ptr(y_op),
            _lib.ptr(y_color), _lib.ptr(g_op), _lib.ptr(g_color), _lib.ptr(g_cov), _lib.ptr(hcat), _lib.ptr(d_src),
            _lib.ptr(d_anchor), _lib.ptr(dz1), _lib.ptr(dz2_op), _lib.ptr(dz2_color), None if defer else _lib.ptr(dW1cat),
            None if defer else _lib.ptr(db1cat), None if defer else _ptr_array(dW2), None if defer else _ptr_array(db2), n,
            int(ctx.tiled), _lib.ptr(ws), ws.numel(), _lib.current_stream()), "cgs_anchor_mlp3_backward_rows")
        grads = [d_src if ctx.needs_input_grad[0] else None, None, d_anchor if ctx.needs_input_grad[2] else None, None, None]
        wgrads = []
        for i in range(3):
            wgrads += [dW1cat[hp * i:hp * i + 50], db1cat[hp * i:hp * i + 50], dW2[i], db2[i]]
        if not defer:
            return tuple(grads + wgrads)
        _defer(_m3_wgrad_job(x, x.shape[1], hcat, dz1, dz2_op, dz2_color, g_cov, dW1cat, db1cat, dW2, db2, n, ws, ctx.params,
                             wgrads))
        return tuple(grads) + (None,) * 12


def anchor_mlp3_rows(feat_src, src_row, anchor_vis, cam_center, mo: nn.Sequential, mc: nn.Sequential, mv: nn.Sequential, pre=None):
    """(mlp_opacity(x), mlp_color(x), mlp_cov(x)) with x[r] = [feat_src[src_row[r]] | unit view vector | distance] of
    anchor_vis[r] seen from cam_center, assembled inside the fused launch.  pre: anchor_mlp3_rows_launch() of the same operands."""
    params = []
    for s in (mo, mc, mv):
        params += [s[0].weight, s[0].bias, s[2].weight, s[2].bias]
    return _AnchorMLP3Rows.apply(feat_src, src_row, anchor_vis, cam_center, pre, *params)


def anchor_mlp3_rows_launch(feat_src, src_row, anchor_vis, cam_center, mo: nn.Sequential, mc: nn.Sequential, mv: nn.Sequential,
                            need_grad=True):
    """The launch of anchor_mlp3_rows() on plain values, ahead of the node (see _AnchorMLP3Rows.launch)."""
    params = []
    for s in (mo, mc, mv):
        params += [s[0].weight, s[0].bias, s[2].weight, s[2].bias]
    return _AnchorMLP3Rows.launch(feat_src, src_row, anchor_vis, cam_center, params, need_grad)


def anchor_mlp3(x, mo: nn.Sequential, mc: nn.Sequential, mv: nn.Sequential):
    """(mlp_opacity(x), mlp_color(x), mlp_cov(x)) in one fused launch each way."""
    params = []
    for s in (mo, mc, mv):
        params += [s[0].weight, s[0].bias, s[2].weight, s[2].bias]
    return _AnchorMLP3.apply(x, *params)
