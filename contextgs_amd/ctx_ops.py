"""Fused element-wise stages of the per-level context model (training path), include/cgs.h
`cgs_rowcat_*`, `cgs_noise_quant_*`, `cgs_level_rate_*` (csrc/ctx.hip).

Reference: scene/gaussian_model.py:1556-1707 (multi_scale_generating).  Each op is an
autograd.Function whose forward and backward are ONE HIP launch each; the torch composition
they replace stays in context_model.py as the eval / unsupported-shape path and as the
parity reference of tests/test_ctx_ops_gpu.py.
"""
from __future__ import annotations

import ctypes as C
import itertools
import threading
import weakref

import os

import torch

from . import _lib

_f32 = torch.float32


def _c(t):
    t = t if t.dtype == _f32 else t.float()
    return t if t.is_contiguous() else t.contiguous()


def _ptrs(ts):
    return (C.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def _ints(v):
    return (C.c_int * len(v))(*v)


class _RowCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, *srcs):
        # spec[s] = (idx LongTensor | None, distinct: bool, rowmask uint8/bool [rows] | None)
        srcs = [_c(s) for s in srcs]
        _lib.require_device(*srcs)
        idxs = [sp[0] for sp in spec]
        masks = ctx.masks = [None if sp[2] is None else (sp[2] if sp[2].dtype == torch.uint8 else sp[2].view(torch.uint8)).contiguous()
                             for sp in spec]
        n = next((i.shape[0] for i in idxs if i is not None), srcs[0].shape[0])
        for s, i in zip(srcs, idxs):
            if i is None and s.shape[0] != n:
                raise ValueError("rowcat: an un-indexed source must have one row per output row")
        widths = [int(s.shape[1]) for s in srcs]
        out = torch.empty(n, sum(widths), dtype=_f32, device=srcs[0].device)
        if n > 0:
            _lib.check(_lib.lib().cgs_rowcat_fwd_masked(len(srcs), _ptrs(srcs), _ptrs(idxs), _ptrs(masks), _ints(widths),
                                                        _ints(widths), n, _lib.ptr(out), _lib.current_stream()),
                       "cgs_rowcat_fwd")
        ctx.spec, ctx.widths, ctx.n = spec, widths, n
        ctx.rows = [int(s.shape[0]) for s in srcs]
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        grads, modes = [], []
        for k, ((idx, distinct, _m), w, rows) in enumerate(zip(ctx.spec, ctx.widths, ctx.rows)):
            if not ctx.needs_input_grad[1 + k]:
                grads.append(None)
                modes.append(0)
            elif idx is None:
                grads.append(torch.empty(rows, w, dtype=_f32, device=g.device))      # every row written
                modes.append(1)
            else:
                # distinct rows that number as many as the source has: every row is written, no zero fill
                full = distinct and ctx.n == rows
                grads.append((torch.empty if full else torch.zeros)(rows, w, dtype=_f32, device=g.device))
                modes.append(1 if distinct else 2)
        if any(modes) and ctx.n > 0:
            _lib.check(_lib.lib().cgs_rowcat_bwd_masked(len(grads), _ptrs(grads), _ptrs([sp[0] for sp in ctx.spec]),
                                                        _ptrs(ctx.masks), _ints(ctx.widths), _ints(ctx.widths), _ints(modes),
                                                        ctx.n, _lib.ptr(g), _lib.current_stream()), "cgs_rowcat_bwd")
        return (None, *grads)


def rowcat(parts):
    """cat([src[idx] for (src, idx, distinct[, rowmask]) in parts], dim=1) in one launch; idx None = all rows in order;
    distinct says the rows of idx do not repeat (plain scatter backward instead of atomics); rowmask (bool / uint8 per
    SOURCE row): the source counts as src * rowmask[:, None]."""
    spec = tuple((p[1], bool(p[2]), p[3] if len(p) > 3 else None) for p in parts)
    return _RowCat.apply(spec, *[p[0] for p in parts])


def gather_rows_nograd(x, idx):
    """x[idx] (rows) through the rowcat kernel, no autograd: for use inside backward passes."""
    x = _c(x)
    n, w = int(idx.shape[0]), int(x[0].numel())
    out = torch.empty((n,) + tuple(x.shape[1:]), dtype=_f32, device=x.device)
    if n > 0:
        _lib.check(_lib.lib().cgs_rowcat_fwd(1, _ptrs([x]), _ptrs([idx]), _ints([w]), _ints([w]), n, _lib.ptr(out),
                                             _lib.current_stream()), "cgs_rowcat_fwd")
    return out


_ROW_STAMPS = {}      # (device, n_full) -> [uint32 stamp tensor, generation]
_ROW_STAMPS_LOCK = threading.Lock()     # autograd runs backward nodes on its own threads


def add_rows_(out, idx, g):
    """out[idx] += g for DISTINCT rows idx (out [N, ...], g [n, ...] fp32): cgs_add_rows on the device, index_add_ elsewhere."""
    n = int(idx.shape[0])
    if n == 0:
        return out
    w = int(out[0].numel())
    if (out.is_cuda and out.dtype == _f32 and g.dtype == _f32 and out.is_contiguous() and idx.dtype == torch.int64
            and idx.is_contiguous() and 1 <= w <= 256 and int(g.numel()) == n * w):
        g = g.contiguous()
        _lib.check(_lib.lib().cgs_add_rows(_lib.ptr(g), _lib.ptr(idx), n, int(out.shape[0]), w, _lib.ptr(out), _lib.current_stream()),
                   "cgs_add_rows")
        return out
    out.view(out.shape[0], -1).index_add_(0, idx, g.reshape(n, -1).to(out.dtype))
    return out


def zero_unlisted_rows(idx, n_full, arrays):
    """Zeros in every row of `arrays` ([n_full, ...] float32, same device) that `idx` (int64, distinct rows < n_full) does not
    name — what torch.zeros gave the gradient buffers of a view's backward before the kernel overwrote the listed rows, without
    filling the 99 % that are overwritten anyway (cgs_mark_rows / cgs_zero_unmarked_rows).  The listed rows stay as they
    are (uninitialised: the caller's kernel writes them).  Every call marks afresh (an index buffer refilled through
    a raw pointer keeps its Python identity and version, so marks can not be reused on that evidence); an index >= n_full is
    ignored by the kernel."""
    if int(n_full) == 0 or not arrays:
        return
    dev = arrays[0].device
    L = _lib.lib()
    key = (str(dev), int(n_full))
    stream = _lib.current_stream()
    with _ROW_STAMPS_LOCK:
        st = _ROW_STAMPS.get(key)
        if st is None or st[1] >= 0xFFFFFFF0:
            if st is None and len(_ROW_STAMPS) >= 8:       # anchor counts change with densification: keep the newest few arrays
                _ROW_STAMPS.pop(next(iter(_ROW_STAMPS)))
            st = _ROW_STAMPS[key] = [torch.zeros(max(int(n_full), 1), dtype=torch.int32, device=dev), 0]
        st[1] += 1
        gen, stamp = st[1], st[0]
        # (enqueued under the lock: two backward threads on one stream must not interleave mark / zero pairs of one stamp array)
        _lib.check(L.cgs_mark_rows(_lib.ptr(idx), int(idx.shape[0]), int(n_full), gen, _lib.ptr(stamp), stream), "cgs_mark_rows")
        for a0 in range(0, len(arrays), 4):
            part = arrays[a0:a0 + 4]
            _lib.check(L.cgs_zero_unmarked_rows(_lib.ptr(stamp), gen, int(n_full), len(part), _ptrs(part),
                                                _ints([int(a[0].numel()) for a in part]), stream), "cgs_zero_unmarked_rows")


_seed_counter = itertools.count(1)


def next_seed() -> int:
    """A fresh 63-bit stream id per call, derived from torch's seed without a device sync (63 bits: the value
    travels through autograd.Function.apply, whose argument recorder (torch.profiler record_shapes) only takes int64)."""
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + next(_seed_counter) * 0xD1B54A32D192ED03) & (2 ** 63 - 1)


class HyperDirect:
    """Where the fused levels' backward kernels leave the gradient of the noisy hyper latents: ONE [N, C] buffer in the latents' own
    row order, row rows[r] written by the level that owns anchor rows[r] (cgs_ctx_level_bwd2) — instead of one strided column slice
    of dX per level that the hyper prior's backward gathers through the inverse coding permutation (a 49 us pass at 1 M anchors).
    Created by EntropyBottleneck.training_step_forms(sizes=...), reachable from every level block as `block._cgs_hyp_direct =
    (holder, block index)`; the hyper prior's backward takes the buffer when every non-empty block was written this way and
    repairs the others' rows otherwise."""

    def __init__(self, n_rows: int, channels: int, sizes):
        self.n, self.c, self.sizes = int(n_rows), int(channels), tuple(int(t) for t in sizes)
        self.buf, self.done = None, set()

    def buffer(self, device):
        if self.buf is None:
            self.buf = torch.empty(self.n, self.c, dtype=_f32, device=device)
        return self.buf

    def take(self):
        buf, done = self.buf, self.done
        self.buf, self.done = None, set()
        return buf, done


class RowSource:
    """The three per-anchor parameter tensors (features [N,D], scaling [N,S], offsets [N,K,3]) read THROUGH a row
    index by the level kernels, instead of being gathered into coding order first.

    Forward: `noise_quant(..., src=self, rows=perm[lo:hi])` reads rows perm[lo:hi] in place.  Backward: every
    level scatters the gradients of its (distinct) rows straight into one full-size buffer per tensor, and the
    node behind `self.token` — which autograd runs after all the levels, because each of them consumed the token —
    hands the three buffers to the parameters.  Compared with gather -> split -> ... -> cat -> index_copy this
    drops a full read+write pass over the tensors in each direction and the per-level gradient concatenation."""

    def __init__(self, feat, scal, off, complete: bool):
        self.shapes = (feat.shape, scal.shape, off.shape)
        self.f, self.s = _c(feat.detach()), _c(scal.detach())
        self.s_orig = scal                      # the tensor the caller passed (rate_model checks WHOSE scaling it is)
        self.o = _c(off.detach()).reshape(off.shape[0], -1)
        _lib.require_device(self.f, self.s, self.o)
        self.complete = bool(complete)          # the levels' rows cover every row: no zero fill needed
        self.rows_read = self.rows_written = 0
        self.grads = None
        # (round 6) ONE anchor-gradient buffer for the levels of a backward: every fused level gathers anchor rows (its own, or
        # its parents'), and as separate [N,3] outputs that was a zero fill per level + autograd's adds (3 fills + 2 adds of
        # 12 MB at 1 M anchors).  The levels that asked for the anchor gradient register in the forward; in the backward the first
        # one to run creates the zeroed buffer, each ADDS its rows, and the last one hands the buffer to autograd.
        self.anchor_users = 0
        self.anchor_acc = None
        self.sums = None            # device double [3]: sums of the values the levels read (RowSource.means())
        self.token = _RowSourceFn.apply(self, feat, scal, off)

    def sums_buffer(self):
        """The accumulator the level kernels add the sums of their source rows to: private to this RowSource (two sources
        alive at once — interleaved forwards, a second stream — never share partial sums), zeroed at creation."""
        if self.sums is None:
            # (+ 16 doubles behind the accumulator: the step's [levels, 3] table of rate sums is carved from the same zero fill,
            #  rate_sum_table)
            nd = int(_lib.lib().cgs_means_accum_doubles())
            self._sums_all = torch.zeros(nd + 16, dtype=torch.float64, device=self.f.device)
            self.sums = self._sums_all[:nd]
            self._table_taken = False
        return self.sums

    def rate_sum_table(self, levels: int):
        """A zeroed float32 [levels, 3] table for the rate node of THIS step (cgs_rate_sub_fwd / cgs_level_rate_fwd add their three
        sums into a row): the tail of the accumulator's allocation, zeroed by the same fill — once per RowSource; None otherwise."""
        if self.sums is None or self._table_taken or 3 * levels > 32:
            return None
        self._table_taken = True
        return self._sums_all[self.sums.shape[0]:].view(_f32)[:3 * levels].view(levels, 3)

    def means(self):
        """float32 [3] = (features.mean(), scaling.mean(), offsets.mean()) once every row has been read by a level:
        the three clamp centres of the rate model (scene/gaussian_model.py:1664-1668) without a pass of their own."""
        n = self.f.shape[0]
        if self.sums is None or self.rows_read != n:
            raise RuntimeError("RowSource.means(): the levels have not read every row")
        out = torch.empty(3, dtype=_f32, device=self.f.device)
        _lib.check(_lib.lib().cgs_means_finalize(_lib.ptr(self.sums), self.f.numel(), self.s.numel(), self.o.numel(),
                                                 _lib.ptr(out), _lib.current_stream()), "cgs_means_finalize")
        return out

    def grad_buffers(self):
        if self.grads is None:
            n, (D, S, O) = self.f.shape[0], (self.f.shape[1], self.s.shape[1], self.o.shape[1])
            flat = (torch.empty if self.complete else torch.zeros)(n * (D + S + O), dtype=_f32, device=self.f.device)
            gf, gs, go = torch.split(flat, [n * D, n * S, n * O])
            self.grads = (gf.view(n, D), gs.view(n, S), go.view(n, O))
        return self.grads


class _RowSourceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, feat, scal, off):
        # (a WEAK reference: `src` keeps this node's output — its token — so a strong one would be a cycle through the C++
        #  node, invisible to Python's collector; the level nodes that consumed the token hold `src` until the graph is freed)
        ctx.src_ref = weakref.ref(src)
        ctx.set_materialize_grads(False)     # the token's gradient carries nothing: no zeros(1) for it
        return feat.new_empty(1)             # (its VALUE is never read either: no fill launch)

    @staticmethod
    def backward(ctx, _g):
        src = ctx.src_ref()
        if src is None:                      # no level ever took the token: nothing was read, nothing to hand on
            return None, None, None, None
        if src.rows_written != src.rows_read:
            raise RuntimeError("RowSource: a level that read rows in the forward did not run its backward "
                               f"({src.rows_written} of {src.rows_read} rows have gradients)")
        gf, gs, go = src.grad_buffers()
        src.grads = None
        need = ctx.needs_input_grad
        return (None, gf.view(src.shapes[0]) if need[1] else None, gs.view(src.shapes[1]) if need[2] else None,
                go.view(src.shapes[2]) if need[3] else None)


class RateSide:
    """Hand-over of one level's rate gradients from _LevelRate.backward to _NoiseQuant.backward.

    The level outputs have two consumers: everything downstream (dense gradients) and the rate of the ~15 % chosen
    rows.  Instead of scattering the rate gradients into N-row zero buffers that autograd then adds to the dense
    ones (a fill and three full-size adds per level), _LevelRate leaves them COMPACT here and returns no gradient;
    the noise_quant backward — which autograd runs after every consumer of its outputs, so after _LevelRate — adds
    row map[r] of them on the fly.  Needs the RowSource path (the kernel that scatters is the one that adds)."""

    def __init__(self):
        self.map = self.f = self.s = self.o = self.q = None
        # the fused rate-subset path (cgs_rate_sub_*, `rate_lazy` levels): forward -> rate node: the level's input rows and
        # mlp_grid's weights; rate node's backward -> level node's backward: the input-row gradient of the mean / scale branch
        # [m, in] and the weight gradients of that branch (dW1, db1, dW2, db2: the level kernel accumulates its own into them)
        self.X = self.weights = self.dx = self.dW = None


class _NoiseQuant(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xf, xs, xo, qadj, seed, q0, outs, src, rows, token, side):
        qadj = _c(qadj)
        if src is not None:
            xf, xs, xo = src.f, src.s, src.o
            n = int(rows.shape[0])
            assert rows.dtype == torch.int64 and rows.is_contiguous()
            src.rows_read += n
        else:
            xf, xs, xo = _c(xf), _c(xs), _c(xo)
            n = xf.shape[0]
        _lib.require_device(xf, xs, xo, qadj)
        ctx.set_materialize_grads(False)     # an output nobody differentiates (Q outside the rate subset) arrives as None, not zeros
        D, S, O = xf.shape[1], xs.shape[1], xo.shape[1]
        if outs is None:
            mk = lambda w: torch.empty(n, w, dtype=_f32, device=qadj.device)
            yf, ys, yo = mk(D), mk(S), mk(O)
        else:           # row slices of caller-owned buffers (the level outputs land side by side: no cat afterwards)
            yf, ys, yo = outs
            assert yf.shape == (n, D) and ys.shape == (n, S) and yo.shape == (n, O)
            assert yf.is_contiguous() and ys.is_contiguous() and yo.is_contiguous()
        Q = torch.empty(n, 3, dtype=_f32, device=qadj.device)
        _lib.check(_lib.lib().cgs_noise_quant_fwd(
            _lib.ptr(xf), _lib.ptr(xs), _lib.ptr(xo), _lib.ptr(qadj), _lib.ptr(rows), n, D, S, O, seed,
            q0[0], q0[1], q0[2], _lib.ptr(yf), _lib.ptr(ys), _lib.ptr(yo), _lib.ptr(Q),
            _lib.ptr(src.sums_buffer()) if src is not None else None, _lib.current_stream()), "cgs_noise_quant_fwd")
        ctx.save_for_backward(qadj, rows)
        ctx.dims, ctx.seed, ctx.q0, ctx.src, ctx.side = (n, D, S, O), seed, q0, src, side
        return yf, ys, yo, Q

    @staticmethod
    def backward(ctx, gf, gs, go, gQ):
        qadj, rows = ctx.saved_tensors
        n, D, S, O = ctx.dims
        src = ctx.src
        gf, gs, go, gQ = (None if t is None else _c(t) for t in (gf, gs, go, gQ))
        dq = torch.empty(n, 3, dtype=_f32, device=qadj.device) if (ctx.needs_input_grad[3] or src is not None) else None
        dx = src.grad_buffers() if src is not None else (None, None, None)
        side = ctx.side if (ctx.side is not None and ctx.side.map is not None and src is not None) else None
        # (a level without a single chosen row has an all -1 map and EMPTY side arrays: nothing to add, and empty tensors
        #  would reach the C entry point as NULL pointers)
        use_side = side is not None and side.f is not None and side.f.numel() > 0
        sd = (side.map, side.f, side.s, side.o, side.q) if use_side else (None,) * 5
        if dq is not None:
            _lib.check(_lib.lib().cgs_noise_quant_bwd(
                _lib.ptr(gf), _lib.ptr(gs), _lib.ptr(go), _lib.ptr(gQ), _lib.ptr(qadj), n, D, S, O, ctx.seed, ctx.q0[0],
                ctx.q0[1], ctx.q0[2], _lib.ptr(dq), _lib.ptr(rows) if src is not None else None, _lib.ptr(dx[0]),
                _lib.ptr(dx[1]), _lib.ptr(dx[2]), *[_lib.ptr(t) for t in sd], _lib.current_stream()), "cgs_noise_quant_bwd")
        if side is not None:
            side.map = side.f = side.s = side.o = side.q = None
        if src is not None:
            src.rows_written += n
            # (no gradient VALUE for the token: it only ties _RowSourceFn into the graph, whose backward hands out the
            #  buffers this kernel filled — a zeros(1) per level cost a fill and an add launch each)
            return None, None, None, dq, None, None, None, None, None, None, None
        return gf, gs, go, dq, None, None, None, None, None, None, None


def noise_quant(xf, xs, xo, qadj, q0, seed=None, outs=None, src=None, rows=None, side=None):
    """(xf + u Qf, xs + u Qs, xo + u Qo, Q[n,3]) with Q = clamp(q0 (1 + tanh(qadj)), 1e-9), u ~ U[-0.5, 0.5).
    outs = (yf, ys, yo) optionally names the (contiguous) tensors to write the three results into.
    src / rows: read x from rows `rows` of a RowSource instead of xf/xs/xo (which are then ignored).
    side: a RateSide that level_rate(..., side=side) of the same level fills in its backward (needs src)."""
    assert side is None or src is not None
    return _NoiseQuant.apply(xf, xs, xo, qadj, next_seed() if seed is None else int(seed),
                             tuple(float(v) for v in q0), outs, src, rows, None if src is None else src.token, side)


class _LevelRate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, yf, ys, yo, Q, pred, loc, masks, grows, x_means, use_clamp, K, side, out):
        yf, ys, yo, Q, pred = _c(yf), _c(ys), _c(yo), _c(Q), _c(pred)
        _lib.require_device(yf, pred)
        n_sub, D = pred.shape[0], yf.shape[1]
        # out: a ZEROED float [3] the caller owns (a row of the step's [levels, 3] table: one fill for all levels)
        if out is not None:     # an independent tensor object on the caller's storage (not an autograd view of it)
            sums = torch.empty(0, dtype=_f32, device=yf.device).set_(out.untyped_storage(), out.storage_offset(), (3,), (1,))
        else:
            sums = torch.zeros(3, dtype=_f32, device=yf.device)
        masks = None if masks is None else _c(masks)
        x_means = None if x_means is None else _c(x_means)
        _lib.check(_lib.lib().cgs_level_rate_fwd(
            _lib.ptr(yf), _lib.ptr(ys), _lib.ptr(yo), _lib.ptr(Q), _lib.ptr(loc), _lib.ptr(pred), _lib.ptr(masks),
            _lib.ptr(grows), _lib.ptr(x_means), int(use_clamp), n_sub, D, K, pred.shape[1], _lib.ptr(sums),
            _lib.current_stream()), "cgs_level_rate_fwd")
        ctx.save_for_backward(yf, ys, yo, Q, pred, loc, masks, grows, x_means)
        ctx.cfg = (int(use_clamp), n_sub, D, K)
        ctx.side = side
        return sums

    @staticmethod
    def backward(ctx, g):
        yf, ys, yo, Q, pred, loc, masks, grows, x_means = ctx.saved_tensors
        use_clamp, n_sub, D, K = ctx.cfg
        n_l = yf.shape[0]
        side = ctx.side if loc is not None else None
        # row-scattered gradients: one zero fill for the four N-row buffers — or, with a RateSide, compact
        # [n_sub, .] arrays (every row written, no fill) that the level's noise_quant backward adds in
        n_o = n_sub if side is not None else n_l
        flat = (torch.empty if side is not None else torch.zeros)(n_o * (D + 6 + 3 * K + 3), dtype=_f32, device=yf.device)
        d_yf, d_ys, d_yo, dQ = torch.split(flat, [n_o * D, n_o * 6, n_o * 3 * K, n_o * 3])
        d_yf, d_ys, d_yo, dQ = d_yf.view(n_o, D), d_ys.view(n_o, 6), d_yo.view(n_o, 3 * K), dQ.view(n_o, 3)
        d_pred = torch.empty_like(pred)
        d_masks = torch.zeros_like(masks) if (masks is not None and ctx.needs_input_grad[6]) else None
        _lib.check(_lib.lib().cgs_level_rate_bwd(
            _lib.ptr(yf), _lib.ptr(ys), _lib.ptr(yo), _lib.ptr(Q), _lib.ptr(loc), _lib.ptr(pred), _lib.ptr(masks),
            _lib.ptr(grows), _lib.ptr(x_means), use_clamp, n_sub, D, K, pred.shape[1], _lib.ptr(_c(g)), _lib.ptr(d_pred),
            _lib.ptr(d_yf), _lib.ptr(d_ys), _lib.ptr(d_yo), _lib.ptr(dQ), _lib.ptr(d_masks), int(side is not None),
            _lib.current_stream()), "cgs_level_rate_bwd")
        if side is not None:
            m = torch.full((n_l,), -1, dtype=torch.int32, device=yf.device)
            m[loc] = torch.arange(n_sub, dtype=torch.int32, device=yf.device)
            side.map, side.f, side.s, side.o, side.q = m, d_yf, d_ys, d_yo, dQ
            return None, None, None, None, d_pred, None, d_masks, None, None, None, None, None, None
        return d_yf, d_ys, d_yo, dQ, d_pred, None, d_masks, None, None, None, None, None, None


def level_rate(yf, ys, yo, Q, pred, loc, masks, grows, x_means, use_clamp, K, side=None, out=None):
    """[bits_feat, bits_scaling, bits_offsets (mask-weighted)] summed over the chosen rows `loc` of a level.
    side: the RateSide given to the noise_quant call that produced yf/ys/yo/Q (their rate gradients then travel
    through it instead of through autograd)."""
    return _LevelRate.apply(yf, ys, yo, Q, pred, loc, masks, grows, x_means, bool(use_clamp), int(K), side, out)


class _RateFinish(torch.autograd.Function):
    """(bit_per_param, bit_per_feat_param, bit_per_scaling_param, bit_per_offsets_param) [4] + the raw numbers of the
    per-level report, from the per-level bit sums and the hyper bit sum: scene/gaussian_model.py:1687-1705 in ONE
    launch each way (cgs_rate_finish_*).  The level rows must be consecutive rows of one [L,3] table."""

    @staticmethod
    def forward(ctx, hsum, cfg, *rows):
        rate, n_f, n_s, n_o, dead = cfg
        L = len(rows)
        base = rows[0]
        for j, r in enumerate(rows):
            if r.data_ptr() != base.data_ptr() + 12 * j:
                raise ValueError("rate_finish: level sums must be consecutive rows of one table")
        out = torch.empty(4, dtype=_f32, device=base.device)
        raw = torch.empty(2 + L, dtype=_f32, device=base.device)
        _lib.check(_lib.lib().cgs_rate_finish_fwd(_lib.ptr(base), L, _lib.ptr(_c(hsum)), float(rate), float(n_f), float(n_s),
                                                  float(n_o), float(dead), _lib.ptr(out), _lib.ptr(raw),
                                                  _lib.current_stream()), "cgs_rate_finish_fwd")
        ctx.cfg, ctx.L = cfg, L
        ctx.mark_non_differentiable(raw)
        return out, raw

    @staticmethod
    def backward(ctx, g, _g_raw):
        rate, n_f, n_s, n_o, _dead = ctx.cfg
        dS = torch.empty(ctx.L, 3, dtype=_f32, device=g.device)
        dh = torch.empty(1, dtype=_f32, device=g.device)
        _lib.check(_lib.lib().cgs_rate_finish_bwd(_lib.ptr(_c(g)), ctx.L, float(rate), float(n_f), float(n_s), float(n_o),
                                                  _lib.ptr(dS), _lib.ptr(dh), _lib.current_stream()), "cgs_rate_finish_bwd")
        return (dh, None, *dS.unbind(0))


def rate_finish(level_sums, hsum, rate, n_feat, n_scaling, n_offsets, dead_frac):
    """level_sums: list of [3] tensors (consecutive rows of one table, see level_rate(out=)); hsum: [1] hyper bit sum.
    Returns (out [4], raw [2+L])."""
    return _RateFinish.apply(hsum, (float(rate), float(n_feat), float(n_scaling), float(n_offsets), float(dead_frac)), *level_sums)


class _RateAll(torch.autograd.Function):
    """The rate of ALL levels plus the scalar tail as one autograd node (scene/gaussian_model.py:1658-1705 on the
    fused path): per level one cgs_level_rate_* launch, then cgs_rate_finish_*.  Compared with one node per level +
    one for the tail: one zero fill for the [levels,3] sum table (forward) and one for the mask-weight gradient of all
    chosen rows (backward), no per-level slice nodes, and the level-row -> chosen-index maps come from the step's
    bookkeeping kernel (ctx_plan.hip) instead of a fill + arange + scatter per level.

    meta: dict(use_clamp, K, finish=(rate, n_f, n_s, n_o, dead), spans=[(lo, hi) rows of `masks` per level],
               sides=[RateSide | None per level], maps=[int32 [n_level] | None per level]);
    masks [T, K]: the mask weights of the chosen rows of all levels, level after level; lv: per level
    (yf, ys, yo, Q, pred, loc).  Returns (out [4], raw [2 + L])."""

    @staticmethod
    def forward(ctx, hsum, masks, x_means, meta, *lv):
        Lc = _lib.lib()
        nl = len(lv) // 6
        K, use_clamp = int(meta["K"]), int(bool(meta["use_clamp"]))
        masks, x_means, hsum = _c(masks), _c(x_means), _c(hsum)
        lv = [t.contiguous() if k % 6 == 5 else _c(t) for k, t in enumerate(lv)]     # every 6th is loc (int64)
        _lib.require_device(masks, x_means, lv[0], lv[4])
        dev = masks.device
        S = meta.get("sum_table")           # (zeroed with the row source's accumulator: RowSource.rate_sum_table)
        if S is None or tuple(S.shape) != (nl, 3):
            S = torch.zeros(nl, 3, dtype=_f32, device=dev)
        stream = _lib.current_stream()
        for j in range(nl):
            yf, ys, yo, Q, pred, loc = lv[6 * j:6 * j + 6]
            lo, hi = meta["spans"][j]
            side = meta["sides"][j]
            if side is not None and side.X is not None:          # a rate_lazy level: MLP branch + rate terms in one launch
                n_sub = int(loc.shape[0])
                assert hi - lo == n_sub
                if n_sub > 0:
                    W1, b1, W2, b2 = side.weights
                    _lib.check(Lc.cgs_rate_sub_fwd(
                        int(side.X.shape[1]), _lib.ptr(side.X), int(side.X.shape[0]), _lib.ptr(loc), n_sub, _lib.ptr(W1), _lib.ptr(b1),
                        _lib.ptr(W2), _lib.ptr(b2), _lib.ptr(yf), _lib.ptr(ys), _lib.ptr(yo), _lib.ptr(Q), masks.data_ptr() + 4 * K * lo,
                        _lib.ptr(x_means), use_clamp, S.data_ptr() + 12 * j, stream), "cgs_rate_sub_fwd")
                continue
            n_sub = int(pred.shape[0])
            assert hi - lo == n_sub and int(loc.shape[0]) == n_sub
            _lib.check(Lc.cgs_level_rate_fwd(
                _lib.ptr(yf), _lib.ptr(ys), _lib.ptr(yo), _lib.ptr(Q), _lib.ptr(loc), _lib.ptr(pred),
                masks.data_ptr() + 4 * K * lo, None, _lib.ptr(x_means), use_clamp, n_sub, yf.shape[1], K, pred.shape[1],
                S.data_ptr() + 12 * j, stream), "cgs_level_rate_fwd")
        rate, n_f, n_s, n_o, dead = meta["finish"]
        out = torch.empty(4, dtype=_f32, device=dev)
        raw = torch.empty(2 + nl, dtype=_f32, device=dev)
        _lib.check(Lc.cgs_rate_finish_fwd(_lib.ptr(S), nl, _lib.ptr(hsum), float(rate), float(n_f), float(n_s), float(n_o),
                                          float(dead), _lib.ptr(out), _lib.ptr(raw), stream), "cgs_rate_finish_fwd")
        ctx.save_for_backward(masks, x_means, *lv)
        ctx.meta = meta
        ctx.mark_non_differentiable(raw)
        # the four entries leave as four outputs: the training loss reads the first only (train.py:207), and selecting it from a [4]
        # output cost a zero fill + a copy in the backward (SelectBackward0); an unread entry now simply has no gradient
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2], out[3], raw

    @staticmethod
    def backward(ctx, g0, g1, g2, g3, _g_raw):
        Lc = _lib.lib()
        masks, x_means = ctx.saved_tensors[:2]
        lv = ctx.saved_tensors[2:]
        meta = ctx.meta
        nl = len(lv) // 6
        K, use_clamp = int(meta["K"]), int(bool(meta["use_clamp"]))
        rate, n_f, n_s, n_o, _dead = meta["finish"]
        dev = masks.device
        stream = _lib.current_stream()
        dS = torch.empty(nl, 3, dtype=_f32, device=dev)
        dh = torch.empty(1, dtype=_f32, device=dev)
        g0, g1, g2, g3 = (None if t is None else _c(t) for t in (g0, g1, g2, g3))
        _lib.check(Lc.cgs_rate_finish_bwd4(_lib.ptr(g0), _lib.ptr(g1), _lib.ptr(g2), _lib.ptr(g3), nl, float(rate), float(n_f),
                                           float(n_s), float(n_o), _lib.ptr(dS), _lib.ptr(dh), stream), "cgs_rate_finish_bwd")
        all_lazy = all(sd is not None and sd.X is not None for sd in meta["sides"])
        # (the level-rate kernels add into it; the fused rate-subset kernels write every row of their span)
        d_masks = (torch.empty_like if all_lazy else torch.zeros_like)(masks) if ctx.needs_input_grad[1] else None
        grads = []
        for j in range(nl):
            yf, ys, yo, Q, pred, loc = lv[6 * j:6 * j + 6]
            lo, _hi = meta["spans"][j]
            n_sub, n_l, D = int(pred.shape[0]), int(yf.shape[0]), int(yf.shape[1])
            side = meta["sides"][j]
            if side is not None and side.X is not None:
                n_sub = int(loc.shape[0])
                grads += [None, None, None, None, None, None]
                if n_sub == 0:      # nothing chosen on this level: no side arrays, the level node zero-fills its weight gradients
                    side.X = side.weights = None
                    continue
                X, (W1, b1, W2, b2) = side.X, side.weights
                in_f = int(X.shape[1])
                flat = torch.empty(n_sub * (D + 6 + 3 * K + 3 + in_f), dtype=_f32, device=dev)
                d_yf, d_ys, d_yo, dQ, dx = torch.split(flat, [n_sub * D, n_sub * 6, n_sub * 3 * K, n_sub * 3, n_sub * in_f])
                wflat = torch.empty(W1.numel() + b1.numel() + W2.numel() + b2.numel(), dtype=_f32, device=dev)
                dW1, db1, dW2, db2 = (t.view(w.shape) for t, w in zip(torch.split(wflat, [W1.numel(), b1.numel(), W2.numel(), b2.numel()]),
                                                                      (W1, b1, W2, b2)))
                ws = torch.empty(int(Lc.cgs_rate_sub_bwd_scratch_bytes(in_f, n_sub)), dtype=torch.uint8, device=dev)
                _lib.check(Lc.cgs_rate_sub_bwd(
                    in_f, _lib.ptr(X), int(X.shape[0]), _lib.ptr(loc), n_sub, _lib.ptr(W1), _lib.ptr(b1), _lib.ptr(W2), _lib.ptr(b2),
                    _lib.ptr(yf), _lib.ptr(ys), _lib.ptr(yo), _lib.ptr(Q), masks.data_ptr() + 4 * K * lo, _lib.ptr(x_means), use_clamp,
                    dS.data_ptr() + 12 * j, _lib.ptr(d_yf), _lib.ptr(d_ys), _lib.ptr(d_yo), _lib.ptr(dQ), _lib.ptr(dx),
                    None if d_masks is None else d_masks.data_ptr() + 4 * K * lo, _lib.ptr(dW1), _lib.ptr(db1), _lib.ptr(dW2),
                    _lib.ptr(db2), _lib.ptr(ws), ws.numel(), stream), "cgs_rate_sub_bwd")
                mp = meta["maps"][j]
                if mp is None:
                    mp = torch.full((n_l,), -1, dtype=torch.int32, device=dev)
                    mp[loc] = torch.arange(n_sub, dtype=torch.int32, device=dev)
                side.map, side.f, side.s, side.o, side.q = mp, d_yf.view(n_sub, D), d_ys.view(n_sub, 6), d_yo.view(n_sub, 3 * K), dQ.view(n_sub, 3)
                side.dx, side.dW = dx.view(n_sub, in_f), (dW1, db1, dW2, db2)
                continue
            n_o_rows = n_sub if side is not None else n_l
            flat = (torch.empty if side is not None else torch.zeros)(n_o_rows * (D + 6 + 3 * K + 3), dtype=_f32, device=dev)
            d_yf, d_ys, d_yo, dQ = torch.split(flat, [n_o_rows * D, n_o_rows * 6, n_o_rows * 3 * K, n_o_rows * 3])
            d_yf, d_ys, d_yo, dQ = d_yf.view(n_o_rows, D), d_ys.view(n_o_rows, 6), d_yo.view(n_o_rows, 3 * K), dQ.view(n_o_rows, 3)
            d_pred = torch.empty_like(pred)
            _lib.check(Lc.cgs_level_rate_bwd(
                _lib.ptr(yf), _lib.ptr(ys), _lib.ptr(yo), _lib.ptr(Q), _lib.ptr(loc), _lib.ptr(pred),
                masks.data_ptr() + 4 * K * lo, None, _lib.ptr(x_means), use_clamp, n_sub, D, K, pred.shape[1],
                dS.data_ptr() + 12 * j, _lib.ptr(d_pred), _lib.ptr(d_yf), _lib.ptr(d_ys), _lib.ptr(d_yo), _lib.ptr(dQ),
                None if d_masks is None else d_masks.data_ptr() + 4 * K * lo, int(side is not None), stream),
                "cgs_level_rate_bwd")
            if side is not None:
                m = meta["maps"][j]
                if m is None:
                    m = torch.full((n_l,), -1, dtype=torch.int32, device=dev)
                    m[loc] = torch.arange(n_sub, dtype=torch.int32, device=dev)
                side.map, side.f, side.s, side.o, side.q = m, d_yf, d_ys, d_yo, dQ
                grads += [None, None, None, None, d_pred, None]
            else:
                grads += [d_yf, d_ys, d_yo, dQ, d_pred, None]
        return (dh, d_masks, None, None, *grads)


def rate_all(hsum, masks, x_means, meta, level_tensors):
    """See _RateAll.  level_tensors: flat list (yf, ys, yo, Q, pred, loc) per level.  Returns ((four 0-dim tensors), raw)."""
    o = _RateAll.apply(hsum, masks, x_means, meta, *level_tensors)
    return o[:4], o[4]


class _CtxAssemble(torch.autograd.Function):
    """[anchor[idx] | base_f[pos] | base_s[pos] | own] with the atomics-free CSR backward (cgs_ctx_gather_bwd)."""

    @staticmethod
    def forward(ctx, anchor, base_f, base_s, own, idx, pos, csr):
        anchor, base_f, base_s, own = _c(anchor), _c(base_f), _c(base_s), _c(own)
        _lib.require_device(anchor, base_f, base_s, own)
        n = idx.shape[0]
        srcs, idxs = [anchor, base_f, base_s, own], [idx, pos, pos, None]
        widths = [int(s.shape[1]) for s in srcs]
        out = torch.empty(n, sum(widths), dtype=_f32, device=anchor.device)
        if n > 0:
            _lib.check(_lib.lib().cgs_rowcat_fwd(4, _ptrs(srcs), _ptrs(idxs), _ints(widths), _ints(widths), n,
                                                 _lib.ptr(out), _lib.current_stream()), "cgs_rowcat_fwd")
        ctx.csr, ctx.widths, ctx.n = csr, widths, n
        ctx.rows = (anchor.shape[0], base_f.shape[0])
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        offs, order, prow = ctx.csr
        wa, DF, DS, WO = ctx.widths
        n_anchor, n_par = ctx.rows
        need = ctx.needs_input_grad
        dev = g.device
        d_anchor = torch.zeros(n_anchor, wa, dtype=_f32, device=dev) if need[0] else None
        d_f = torch.empty(n_par, DF, dtype=_f32, device=dev) if need[1] else None
        d_s = torch.empty(n_par, DS, dtype=_f32, device=dev) if need[2] else None
        d_own = None
        if ctx.n > 0:
            if d_anchor is not None or d_f is not None or d_s is not None:
                _lib.check(_lib.lib().cgs_ctx_gather_bwd(_lib.ptr(g), g.shape[1], n_par, _lib.ptr(offs), _lib.ptr(order),
                                                         _lib.ptr(prow), _lib.ptr(d_anchor), _lib.ptr(d_f), _lib.ptr(d_s),
                                                         wa, DF, DS, _lib.current_stream()), "cgs_ctx_gather_bwd")
            if need[3]:
                d_own = g[:, wa + DF + DS:]          # a column slice: its consumer (the split's cat) takes strided rows
        else:
            d_f = None if d_f is None else d_f.zero_()
            d_s = None if d_s is None else d_s.zero_()
            d_own = torch.zeros(0, WO, dtype=_f32, device=dev) if need[3] else None
        return d_anchor, d_f, d_s, d_own, None, None, None


def ctx_assemble(anchor, base_f, base_s, own, idx, pos, csr):
    """The MLP input rows of a level: parent anchor / coded feature / coded scaling rows + the level's own hyper
    rows.  csr = (offs[n_parents+1], order[n_children], parent_row[n_parents]) lists every parent's children."""
    return _CtxAssemble.apply(anchor, base_f, base_s, own, idx, pos, csr)


class _MaskSTE(torch.autograd.Function):
    """get_mask / get_mask_anchor of the model in one launch each way (cgs_mask_ste_*, csrc/mask.hip)."""

    @staticmethod
    def forward(ctx, logits):
        x = _c(logits)
        _lib.require_device(x)
        n, K = x.shape[0], x[0].numel() if x.shape[0] else 1
        mask = torch.empty_like(x)
        alive = torch.empty(n, dtype=torch.bool, device=x.device)
        _lib.check(_lib.lib().cgs_mask_ste_fwd(_lib.ptr(x), n, K, _lib.ptr(mask), _lib.ptr(alive), _lib.current_stream()),
                   "cgs_mask_ste_fwd")
        ctx.save_for_backward(x)
        ctx.mark_non_differentiable(alive)
        ctx.set_materialize_grads(False)       # no zeros [N] for the gradient of the (bool) `alive` output
        return mask, alive

    @staticmethod
    def backward(ctx, g, _g_alive):
        (x,) = ctx.saved_tensors
        if g is None:
            return None
        d = torch.empty_like(x)
        _lib.check(_lib.lib().cgs_mask_ste_bwd(_lib.ptr(x), _lib.ptr(_c(g)), x.numel(), _lib.ptr(d), _lib.current_stream()),
                   "cgs_mask_ste_bwd")
        return d


class _MaskSTEAttach(torch.autograd.Function):
    """The autograd node of mask_ste() created LATER than its values: `values` = the mask mask_ste_values() computed earlier
    in the forward.  The autograd engine runs ready nodes in the reverse order of their creation; a node created at the top of
    render() (where the anchor mask is needed for the level plan) is run after every node of the context model, so `_mask`'s
    gradient — complete once the expansion's and the rate model's backward are done — became final only after the level kernels
    and its all-reduce (dist.GradientSync) could not start beside them."""

    @staticmethod
    def forward(ctx, logits, values):
        ctx.save_for_backward(logits.detach())
        ctx.set_materialize_grads(False)
        return values.view_as(values)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        if g is None:
            return None, None
        x = _c(x)
        d = torch.empty_like(x)
        _lib.check(_lib.lib().cgs_mask_ste_bwd(_lib.ptr(x), _lib.ptr(_c(g)), x.numel(), _lib.ptr(d), _lib.current_stream()),
                   "cgs_mask_ste_bwd")
        return d, None


def mask_ste_values(logits):
    """(mask, any_alive) of mask_ste() WITHOUT an autograd node (attach one later: mask_ste_attach)."""
    with torch.no_grad():
        m, alive = _MaskSTE.apply(logits.detach())
    return m, alive


def mask_ste_attach(logits, values):
    return _MaskSTEAttach.apply(logits, values)


def mask_ste(logits):
    """(mask, any_alive): mask = ((sigmoid(m) > 0.01) - sigmoid(m)) + sigmoid(m) with the sigmoid's gradient
    (scene/gaussian_model.py:295-299), any_alive[a] = some offset of anchor a has mask > 0 (:302-310)."""
    return _MaskSTE.apply(logits)


_means3_ws = {}


def means3(a, b, c, exp_b=False):
    """float32 [3] = (a.mean(), (exp(b) if exp_b else b).mean(), c.mean()) in one launch, no autograd (the clamp
    centres of the rate model, scene/gaussian_model.py:1664-1668, as the fused rate kernel consumes them)."""
    a, b, c = _c(a.detach()), _c(b.detach()), _c(c.detach())
    _lib.require_device(a, b, c)
    L = _lib.lib()
    ws = _means3_ws.get(a.device)
    if ws is None:
        ws = _means3_ws[a.device] = torch.empty(L.cgs_means3_scratch_bytes(), dtype=torch.uint8, device=a.device)
    out = torch.empty(3, dtype=_f32, device=a.device)
    _lib.check(L.cgs_means3(_lib.ptr(a), a.numel(), _lib.ptr(b), b.numel(), int(bool(exp_b)), _lib.ptr(c), c.numel(),
                            _lib.ptr(ws), ws.numel(), _lib.ptr(out), _lib.current_stream()), "cgs_means3")
    return out


_PINNED_FREE = []
CHOOSE_SLOTS = int(os.environ.get("CGS_CHOOSE_SLOTS", "32"))          # A/B knob: 1 = one `meta` array for all blocks (rounds 4-6)


def _pinned_i32(k):
    """A small pinned int32 buffer from a free list (allocating pinned memory per step would cost more than the read)."""
    while _PINNED_FREE:
        b = _PINNED_FREE.pop()
        if b.numel() >= k:
            return b
    return torch.empty(max(k, 16), dtype=torch.int32, pin_memory=True)


def choose_rows_begin(perm, n, mask, given, seed, thresh, anchor, anchor_ref, mask_ref, bounds):
    """First half of choose_rows: launches the flag / count kernel and returns a handle WITHOUT waiting for it, so that
    the caller can enqueue more work (or reach a synchronisation it has to make anyway) before the counts are read."""
    L = _lib.lib()
    dev = (perm if perm is not None else anchor if anchor is not None else mask).device
    as_u8 = lambda t: None if t is None else (t if t.dtype == torch.uint8 else t.view(torch.uint8)).contiguous()
    mask_u8, given_u8, mref_u8 = as_u8(mask), as_u8(given), as_u8(mask_ref)
    nlev = len(bounds) - 1
    nblk = int(L.cgs_ctx_choose_blocks(n))
    flags = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    counts = torch.empty(max(nblk, 1), dtype=torch.int32, device=dev)
    # the blocks' closing atomics spread over CHOOSE_SLOTS lines (one counter for ~1000 blocks serialises on its address); the host
    # adds the slots up after the read it makes anyway
    slots, sw = (CHOOSE_SLOTS, int(L.cgs_ctx_choose_slot_ints())) if CHOOSE_SLOTS >= 2 else (1, 2 + nlev)
    meta = torch.empty(slots * sw, dtype=torch.int32, device=dev)
    b_host = (C.c_int64 * (nlev + 1))(*[int(b) for b in bounds])
    args = (_lib.ptr(perm), n, _lib.ptr(mask_u8), _lib.ptr(given_u8), int(seed), float(thresh),
            _lib.ptr(None if anchor is None else _c(anchor)), _lib.ptr(None if anchor_ref is None else _c(anchor_ref)), _lib.ptr(mref_u8),
            b_host, nlev, _lib.ptr(flags), _lib.ptr(counts), _lib.ptr(meta))
    if slots >= 2:
        _lib.check(L.cgs_ctx_choose_flags_slots(*args, slots, _lib.current_stream()), "cgs_ctx_choose_flags_slots")
    else:
        _lib.check(L.cgs_ctx_choose_flags(*args, _lib.current_stream()), "cgs_ctx_choose_flags")
    # the counts travel to a pinned host buffer behind the kernel, with an event of their own: choose_rows_end waits for
    # THAT copy (long finished by then) instead of draining whatever the caller has queued since
    pinned = _pinned_i32(slots * sw)
    pinned[:slots * sw].copy_(meta, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return dict(perm=perm, n=n, nlev=nlev, flags=flags, counts=counts, meta=meta, b_host=b_host, dev=dev,
                keep=(mask_u8, given_u8, mref_u8), pinned=pinned, event=ev, slots=(slots, sw))


def choose_rows_end(h):
    """Second half: read the counts (the step's synchronisation of the context model), compact the chosen rows."""
    L = _lib.lib()
    h["event"].synchronize()
    slots, sw = h["slots"]
    k_ = 2 + h["nlev"]
    if slots >= 2:
        part = h["pinned"][:slots * sw].view(slots, sw)[:, :k_]
        host = [int(bool(part[:, 0].any()))] + [int(v) for v in part[:, 1:].sum(dim=0).tolist()]
    else:
        host = h["pinned"][:k_].tolist()
    _PINNED_FREE.append(h.pop("pinned"))
    stale, live, per_level = bool(host[0]), int(host[1]), [int(v) for v in host[2:]]
    total = sum(per_level)
    dev = h["dev"]
    nz = torch.empty(total, dtype=torch.int64, device=dev)
    rows = torch.empty(total, dtype=torch.int64, device=dev)
    loc = torch.empty(total, dtype=torch.int64, device=dev)
    sub_map = None
    if total > 0 and not stale:
        sub_map = torch.empty(max(h["n"], 1), dtype=torch.int32, device=dev)
        c_host = (C.c_int64 * h["nlev"])(*per_level)
        _lib.check(L.cgs_ctx_choose_compact(_lib.ptr(h["flags"]), _lib.ptr(h["counts"]), _lib.ptr(h["perm"]), h["n"],
                                            h["b_host"], h["nlev"], _lib.ptr(nz), _lib.ptr(rows), _lib.ptr(loc),
                                            _lib.ptr(sub_map), c_host, _lib.current_stream()), "cgs_ctx_choose_compact")
    return stale, live, per_level, nz, rows, loc, sub_map


def choose_rows(perm, n, mask, given, seed, thresh, anchor, anchor_ref, mask_ref, bounds):
    """The rate subset of one training step in coding order (csrc/ctx_plan.hip): two launches and ONE host read.

    perm: int64 [n] coding-order permutation (None = identity); mask / given / mask_ref: bool [n] or None; anchor /
    anchor_ref: float32 [n,3] or None (plan-validity check); bounds: python list of the level boundaries in coding
    order.  Returns (stale, live_count, per-level counts, nz, rows, loc, sub_map) — nz / rows / loc hold, for every
    chosen row in coding order, its coding-order position, original anchor index and level-local position; sub_map
    (int32 [n], coding order) is the inverse per level: a row's index in its level's part of the chosen list or -1
    (None when nothing was compacted); `stale` says the
    anchors / mask no longer equal the reference copies (the caller rebuilds its plan and calls again)."""
    return choose_rows_end(choose_rows_begin(perm, n, mask, given, seed, thresh, anchor, anchor_ref, mask_ref, bounds))


# ---- one launch per level and direction for the every-row half of the level loop (csrc/ctx_level.hip) ---------------
ANCHOR_SHARED = os.environ.get("CGS_ANCHOR_SHARED", "1") != "0"      # A/B knob: 0 = one zero-filled [N,3] anchor gradient per level (rounds 5)
HYPER_DIRECT = os.environ.get("CGS_HYPER_DIRECT", "1") != "0"        # A/B knob: 0 = the hyper latents' gradient leaves the levels as column slices of dX (gather_rows_segmented)
PREFIX_INPLACE = os.environ.get("CGS_PREFIX_INPLACE", "1") != "0"    # A/B knob: 0 = the coded prefix's gradient as its own tensors, summed by autograd


_JOINED_GRADS = {}       # data pointer -> (rows, width) of the gradients _JoinRows' backward split this step (cleared per forward)


def note_joined_grad(g):
    """_JoinRows.backward: `g` [rows, w] is the gradient of level outputs lying side by side in coding order."""
    if g is not None and g.dim() == 2 and g.dtype == _f32 and g.is_contiguous() and g.is_cuda:
        _JOINED_GRADS[g.data_ptr()] = (int(g.shape[0]), int(g.shape[1]))


def _grad_prefix(part, n_prefix):
    """`part` = rows [n_prefix, n_prefix + n) of a gradient buffer _JoinRows' backward split (note_joined_grad) -> that buffer's
    first n_prefix rows as a view, else None.

    The coded prefix a context level reads IS the earlier levels' outputs, and their gradient from the joined tensor lies in
    the same buffer right in front of this level's rows: the level's backward adds the parents' sums THERE
    (cgs_ctx_gather_bwd_acc bits 1 / 2) and returns no gradient for the prefix.  Autograd still runs the earlier levels'
    nodes after this one (the edges exist), they find the sums in their own slice.  Before: a [prefix, 50] + [prefix, 6]
    tensor per context level and six add launches by the engine per step (profiles/r06_launch_attribution.txt)."""
    if part is None or n_prefix <= 0 or part.dim() != 2 or part.dtype != _f32 or not part.is_contiguous() or part.requires_grad:
        return None
    w = int(part.shape[1])
    rec = _JOINED_GRADS.get(part.data_ptr() - 4 * w * n_prefix)
    if rec is None or rec[1] != w or rec[0] < n_prefix + int(part.shape[0]) or part.storage_offset() < w * n_prefix:
        return None
    return torch.as_strided(part, (n_prefix, w), (w, 1), part.storage_offset() - w * n_prefix)
_LEVEL_DIMS = (50, 6, 30, 12, 100, 175)      # features, scaling, offsets, hyper, hidden, outputs of mlp_grid: the kernels' instance


def level_fused_supported(seq, anchor, hyp, src) -> bool:
    """Shapes / layouts cgs_ctx_level_* are instantiated for: the reference's dimensions (feat 50, scaling 6, 10 offsets,
    hyper 12, mlp_grid {71|15} -> 100 -> 175) read through a RowSource."""
    try:
        l1, l2 = seq[0], seq[2]
        ok = (len(seq) == 3 and l1.out_features == 100 and l2.out_features == 175 and l1.in_features in (71, 15))
    except Exception:
        return False
    return bool(ok and src is not None and src.f.shape[1] == 50 and src.s.shape[1] == 6 and src.o.shape[1] == 30
                and hyp.dim() == 2 and hyp.shape[1] == 12 and hyp.dtype == _f32 and hyp.is_cuda
                and anchor.dtype == _f32 and anchor.is_cuda and anchor.dim() == 2 and anchor.shape[1] == 3)


class _LevelFused(torch.autograd.Function):
    """One level of the training level loop (scene/gaussian_model.py:1594-1616) as ONE node and one launch per direction for
    the work that runs on every row: input-row assembly (:1596-1599 / :1711-1724), mlp_grid's hidden layer and three step-size
    outputs (:1600-1608), the noisy values of the level's rows of the three parameter tensors (:1610-1616) — cgs_ctx_level_fwd /
    cgs_ctx_level_bwd (weight gradients of that branch inside the backward kernel) — plus the mean / scale outputs on the rate
    subset `loc` (the same fused MLP kernels as mlp._LevelMLP, on rows of the X the level kernel wrote).

    cfg: dict(a_rows, a_mask, pos, csr, loc, n_stat, src, rows, seed, q0, outs, side) — see level_fused()."""

    @staticmethod
    def launch(anchor, base_f, base_s, hyp, W1, b1, W2, b2, cfg):
        """The forward's launch on plain values (no autograd): (X, Q, (W1c, b1c, W2c, b2c)).  Called by forward(), or EARLIER by
        level_fused_launch() — the level kernels of a step need nothing the host has to wait for, so context_model enqueues them in
        front of the rate subset's read-back and hands the result to forward() through cfg["pre"]."""
        from . import mlp as _mlp
        L = _lib.lib()
        _mlp._drop_stale_deferred()
        src, rows = cfg["src"], cfg["rows"]
        anchor_c, hyp_c = _c(anchor.detach()), _c(hyp.detach())
        W1c, b1c, W2c, b2c = (t.detach().contiguous() for t in (W1, b1, W2, b2))
        n, in_f = int(rows.shape[0]), int(W1c.shape[1])
        hid, out, n_stat = int(W1c.shape[0]), int(W2c.shape[0]), int(cfg["n_stat"])
        assert (hid, out, out - n_stat) == (100, 175, 3) and rows.dtype == torch.int64 and rows.is_contiguous()
        ctxlevel = base_f is not None
        assert in_f == (71 if ctxlevel else 15) and hyp_c.shape == (n, 12)
        bf = _c(base_f.detach()) if ctxlevel else None
        bs = _c(base_s.detach()) if ctxlevel else None
        a_rows, pos, a_mask = cfg["a_rows"], cfg["pos"], cfg["a_mask"]
        assert a_rows.dtype == torch.int64 and a_rows.is_contiguous() and int(a_rows.shape[0]) == n
        if a_mask is not None:
            a_mask = (a_mask if a_mask.dtype == torch.uint8 else a_mask.view(torch.uint8)).contiguous()
        _lib.require_device(anchor_c, hyp_c, W1c, W2c, src.f)
        dev = anchor_c.device
        src.rows_read += n
        yf, ys, yo = cfg["outs"]
        assert yf.shape == (n, 50) and ys.shape == (n, 6) and yo.shape == (n, 30)
        assert yf.is_contiguous() and ys.is_contiguous() and yo.is_contiguous()
        X = torch.empty(n, in_f, dtype=_f32, device=dev)
        Q = torch.empty(n, 3, dtype=_f32, device=dev)
        stream = _lib.current_stream()
        seed, q0 = int(cfg["seed"]), cfg["q0"]
        _lib.check(L.cgs_ctx_level_fwd(
            in_f, _lib.ptr(anchor_c), int(anchor_c.shape[0]), _lib.ptr(a_rows), _lib.ptr(a_mask), _lib.ptr(bf), _lib.ptr(bs),
            int(bf.shape[0]) if ctxlevel else 0, _lib.ptr(pos), _lib.ptr(hyp_c), n, _lib.ptr(W1c), _lib.ptr(b1c), W2c.data_ptr() + 4 * n_stat * hid, b2c.data_ptr() + 4 * n_stat, _lib.ptr(src.f),
            _lib.ptr(src.s), _lib.ptr(src.o), _lib.ptr(rows), seed, q0[0], q0[1], q0[2], _lib.ptr(X), _lib.ptr(yf), _lib.ptr(ys),
            _lib.ptr(yo), _lib.ptr(Q), _lib.ptr(src.sums_buffer()), stream), "cgs_ctx_level_fwd")
        return X, Q, (W1c, b1c, W2c, b2c)

    @staticmethod
    def forward(ctx, anchor, base_f, base_s, hyp, W1, b1, W2, b2, _token, cfg):
        from . import mlp as _mlp
        L = _lib.lib()
        src, rows, loc = cfg["src"], cfg["rows"], cfg["loc"]
        pre = cfg.get("pre")
        X, Q, (W1c, b1c, W2c, b2c) = pre if pre is not None else _LevelFused.launch(anchor, base_f, base_s, hyp, W1, b1, W2, b2, cfg)
        n, in_f = int(rows.shape[0]), int(W1c.shape[1])
        hid, out, n_stat = int(W1c.shape[0]), int(W2c.shape[0]), int(cfg["n_stat"])
        ctxlevel = base_f is not None
        dev = X.device
        yf, ys, yo = cfg["outs"]
        stream = _lib.current_stream()
        pred = x_sub = h_sub = None
        m = 0
        lazy = bool(cfg.get("rate_lazy")) and loc is not None and cfg["side"] is not None
        if lazy:
            # the mean / scale outputs of the chosen rows are formed inside the rate kernels (cgs_rate_sub_*): nothing here
            m = int(loc.shape[0])
            cfg["side"].X, cfg["side"].weights = X, (W1c, b1c, W2c, b2c)
        elif loc is not None:
            m = int(loc.shape[0])
            x_sub = gather_rows_nograd(X, loc)
            pred = torch.empty(m, out, dtype=_f32, device=dev)
            h_sub = torch.empty(m, hid, dtype=_f32, device=dev)
            _lib.check(L.cgs_mlp2_forward(in_f, hid, out, 0, _lib.ptr(x_sub), in_f, _lib.ptr(W1c), _lib.ptr(b1c), _lib.ptr(W2c),
                                          _lib.ptr(b2c), _lib.ptr(pred), out, _lib.ptr(h_sub), m, stream), "cgs_mlp2_forward")
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(X, x_sub, h_sub, W1c, b1c, W2c, b2c)
        # (without "outs": they are this node's OUTPUTS — an output kept on its own ctx is a reference cycle through the C++
        #  node that Python's collector cannot see: every step's level nodes, RowSource and ~350 MB of level tensors stayed
        #  alive for the life of the process, found in round 5 when the device filled up — tools/leak_probe.py)
        ctx.cfg, ctx.dims = {k: v for k, v in cfg.items() if k not in ("outs", "pre")}, (n, in_f, hid, out, n_stat, m)
        ctx.lazy = lazy
        ctx.shapes = (tuple(anchor.shape), None if base_f is None else tuple(base_f.shape), None if base_s is None else tuple(base_s.shape))
        ctx.share_anchor = bool(ANCHOR_SHARED and ctx.needs_input_grad[0])
        if ctx.share_anchor:
            src.anchor_users += 1
        if pred is None:
            pred = torch.empty(0, out, dtype=_f32, device=dev)
        return yf, ys, yo, Q, pred

    @staticmethod
    def backward(ctx, gf, gs, go, gQ, d_pred):
        from . import mlp as _mlp
        L = _lib.lib()
        X, x_sub, h_sub, W1, b1, W2, b2 = ctx.saved_tensors
        cfg = ctx.cfg
        n, in_f, hid, out, n_stat, m = ctx.dims
        src, rows, loc = cfg["src"], cfg["rows"], cfg["loc"]
        dev = X.device
        stream = _lib.current_stream()
        gf, gs, go, gQ = (None if t is None else _c(t) for t in (gf, gs, go, gQ))
        side0 = cfg["side"]
        dx_sub = None
        if ctx.lazy and side0 is not None and side0.dW is not None:
            # the rate node's backward (cgs_rate_sub_bwd) already ran: its weight gradients are the start of this module's
            # (no zero fill), its input-row gradient rides into the level kernel as compact rows
            (dW1, db1, dW2, db2), dx_sub = side0.dW, side0.dx
            side0.dW = side0.dx = side0.X = side0.weights = None
        else:
            dW1, db1, dW2, db2 = _mlp._zeros_views(dev, (hid, in_f), (hid,), (out, hid), (out,))      # one fill for the module
        ws = None if ctx.lazy else _mlp._wgrad_workspace(dev)
        # the rate subset's branch first: its input gradient rides into the level kernel as compact rows
        if (not ctx.lazy) and d_pred is not None and m > 0:
            d_pred = _c(d_pred)
            dz1s = torch.empty(m, hid, dtype=_f32, device=dev)
            dx_sub = torch.empty(m, in_f, dtype=_f32, device=dev)
            _lib.check(L.cgs_mlp2_backward(in_f, hid, out, 0, _lib.ptr(x_sub), in_f, _lib.ptr(W1), None, _lib.ptr(W2), None,
                                           _lib.ptr(d_pred), out, _lib.ptr(h_sub), _lib.ptr(dx_sub), in_f, 0, _lib.ptr(dz1s), None,
                                           _lib.ptr(dW1), _lib.ptr(db1), _lib.ptr(dW2), _lib.ptr(db2), m, _lib.ptr(ws), ws.numel(),
                                           stream), "cgs_mlp2_backward")
        side = cfg["side"] if (cfg["side"] is not None and cfg["side"].map is not None) else None
        use_side = side is not None and side.f is not None and side.f.numel() > 0
        smap = None
        if use_side:
            smap, sd = side.map, (side.f, side.s, side.o, side.q)
        elif dx_sub is not None and not ctx.lazy:
            # no rate side (gradients of y / Q came through autograd) but a subset branch: its rows still need a map
            smap = torch.full((n,), -1, dtype=torch.int32, device=dev)
            smap[loc] = torch.arange(m, dtype=torch.int32, device=dev)
            z = torch.zeros(m, 50 + 6 + 30 + 3, dtype=_f32, device=dev)
            sd = tuple(t.contiguous() for t in torch.split(z, [50, 6, 30, 3], dim=1))
        else:
            sd = (None,) * 4
        dxf, dxs, dxo = src.grad_buffers()
        dX = torch.empty(n, in_f, dtype=_f32, device=dev)
        ws2 = torch.empty(int(L.cgs_ctx_level_bwd_scratch_bytes()), dtype=torch.uint8, device=dev)
        # the hyper latents' gradient rows straight into the hyper prior's buffer (HyperDirect), when this level's block of latents
        # came from training_step_forms(sizes=) and the latents are indexed like the per-anchor tensors (rows = coding permutation)
        hd, d_hyp_rows = cfg.get("hyp_direct"), None
        if hd is not None and ctx.needs_input_grad[3] and hd[0].n == int(dxf.shape[0]) and hd[0].c == 12 and hd[0].sizes[hd[1]] == n:
            d_hyp_rows = hd[0].buffer(dev)
        _lib.check(L.cgs_ctx_level_bwd2(
            in_f, _lib.ptr(X), _lib.ptr(W1), _lib.ptr(b1), W2.data_ptr() + 4 * n_stat * hid, b2.data_ptr() + 4 * n_stat,
            _lib.ptr(gf), _lib.ptr(gs), _lib.ptr(go), _lib.ptr(gQ), n, int(cfg["seed"]), cfg["q0"][0], cfg["q0"][1], cfg["q0"][2],
            _lib.ptr(rows), int(dxf.shape[0]), _lib.ptr(dxf), _lib.ptr(dxs), _lib.ptr(dxo), _lib.ptr(smap),
            int(sd[0].shape[0]) if sd[0] is not None else 0, *[_lib.ptr(t) for t in sd],
            _lib.ptr(dx_sub), _lib.ptr(dX), _lib.ptr(d_hyp_rows), _lib.ptr(dW1), _lib.ptr(db1), dW2.data_ptr() + 4 * n_stat * hid,
            db2.data_ptr() + 4 * n_stat, _lib.ptr(ws2), ws2.numel(), stream), "cgs_ctx_level_bwd")
        if d_hyp_rows is not None:
            hd[0].done.add(hd[1])
        if side is not None:
            side.map = side.f = side.s = side.o = side.q = None
        src.rows_written += n
        need = ctx.needs_input_grad
        a_shape, f_shape, s_shape = ctx.shapes
        shared = ctx.share_anchor and need[0]
        if shared:
            if src.anchor_acc is None:
                src.anchor_acc = torch.zeros(a_shape, dtype=_f32, device=dev)
            d_anchor = src.anchor_acc
        else:
            d_anchor = torch.zeros(a_shape, dtype=_f32, device=dev) if need[0] else None
        d_f = d_s = None
        if f_shape is not None:                  # context level: the parents' rows, summed over their children without atomics
            offs, order, prow = cfg["csr"]
            # the prefix rows' gradient buffers, when this level's own gradients are row slices right behind them (_grad_prefix)
            in_f_ = _grad_prefix(gf, int(f_shape[0])) if (PREFIX_INPLACE and need[1] and f_shape[1] == 50) else None
            in_s_ = _grad_prefix(gs, int(s_shape[0])) if (PREFIX_INPLACE and need[2] and s_shape[1] == 6) else None
            d_f = in_f_ if in_f_ is not None else (torch.empty(f_shape, dtype=_f32, device=dev) if need[1] else None)
            d_s = in_s_ if in_s_ is not None else (torch.empty(s_shape, dtype=_f32, device=dev) if need[2] else None)
            if d_anchor is not None or d_f is not None or d_s is not None:
                acc = int(shared) | (2 if in_f_ is not None else 0) | (4 if in_s_ is not None else 0)
                _lib.check(L.cgs_ctx_gather_bwd_acc(_lib.ptr(dX), in_f, int(f_shape[0]), _lib.ptr(offs), _lib.ptr(order), _lib.ptr(prow),
                                                    _lib.ptr(d_anchor), _lib.ptr(d_f), _lib.ptr(d_s), 3, 50, 6, acc, stream),
                           "cgs_ctx_gather_bwd")
            if in_f_ is not None:
                d_f = None                  # (already summed into the earlier levels' slices)
            if in_s_ is not None:
                d_s = None
            d_hyp = dX[:, 59:] if (need[3] and d_hyp_rows is None) else None          # a column slice: its consumer takes strided rows
        else:                                    # first level: X = [anchor[a_rows] * mask | hyper]
            d_hyp = dX[:, 3:] if (need[3] and d_hyp_rows is None) else None
            if d_anchor is not None:
                a_mask = cfg["a_mask"]
                if a_mask is not None:
                    a_mask = (a_mask if a_mask.dtype == torch.uint8 else a_mask.view(torch.uint8)).contiguous()
                # (the hyper columns of dX leave as the strided slice above: mode 0 = no store for that source)
                # (shared buffer: these rows may carry an earlier level's sums — mode 2 adds; the rows are distinct, the atomics uncontended)
                _lib.check(L.cgs_rowcat_bwd_masked(2, _ptrs([d_anchor, None]), _ptrs([cfg["a_rows"], None]), _ptrs([a_mask, None]),
                                                   _ints([3, 12]), _ints([3, 12]), _ints([2 if shared else 1, 0]), n, _lib.ptr(dX), stream),
                           "cgs_rowcat_bwd")
        if shared:
            src.anchor_users -= 1
            if src.anchor_users > 0:
                d_anchor = None                 # a later level of this backward hands the shared buffer on
            else:
                src.anchor_acc = None
        return d_anchor, d_f, d_s, d_hyp, dW1, db1, dW2, db2, None, None


def level_fused_launch(anchor, base_f, base_s, hyp, seq, n_stat, a_rows, a_mask, pos, src, rows, outs, q0, seed):
    """The launch of level_fused() on plain values, for a caller that creates the node later: returns `pre` for level_fused(pre=)
    (same arguments, same seed).  base_f / base_s: the coded prefix's VALUES (the buffers the earlier levels' launches wrote)."""
    l1, l2 = seq[0], seq[2]
    cfg = dict(a_rows=a_rows, a_mask=a_mask, pos=pos, n_stat=int(n_stat), src=src, rows=rows, seed=int(seed),
               q0=tuple(float(v) for v in q0), outs=outs)
    return _LevelFused.launch(anchor, base_f, base_s, hyp, l1.weight, l1.bias, l2.weight, l2.bias, cfg)


def level_fused(anchor, base_f, base_s, hyp, seq, n_stat, loc, a_rows, a_mask, pos, csr, src, rows, outs, side, q0, seed=None,
                rate_lazy=False, pre=None):
    """One level of the training level loop.  anchor [N,3]; base_f / base_s: the coded prefix (None for the first level);
    hyp [n,12] the level's noisy hyper latents; seq = mlp_grid[level]; loc: level rows of the rate subset (or None);
    a_rows [n]: anchor row of every level row; a_mask: bool [N] (first level from level 1 up) or None; pos [n]: the parents'
    prefix positions; csr: their children lists (cgs_ctx_gather_bwd); src / rows: RowSource and the level's slice of the coding
    permutation; outs = (yf, ys, yo) slices to write; side: RateSide or None.
    rate_lazy (needs side): the mean / scale outputs of the chosen rows are NOT formed here — rate_all() runs them inside
    the fused rate kernels (cgs_rate_sub_*) from the X this node leaves on `side`; pred is then an empty tensor.
    Returns (yf, ys, yo, Q [n,3], pred [len(loc),175] or an empty tensor)."""
    l1, l2 = seq[0], seq[2]
    cfg = dict(a_rows=a_rows, a_mask=a_mask, pos=pos, csr=csr, loc=loc, n_stat=int(n_stat), src=src, rows=rows,
               seed=next_seed() if seed is None else int(seed), q0=tuple(float(v) for v in q0), outs=outs, side=side,
               rate_lazy=bool(rate_lazy), pre=pre, hyp_direct=getattr(hyp, "_cgs_hyp_direct", None) if HYPER_DIRECT else None)
    return _LevelFused.apply(anchor, base_f, base_s, hyp, l1.weight, l1.bias, l2.weight, l2.bias, src.token, cfg)
