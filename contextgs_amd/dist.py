"""Data-parallel sharding of the hot path over the GPUs of one node.

The reference is single-process / single-GPU (no torch.distributed anywhere);
this is new work, designed for MI355X: one process per GPU, RCCL (`backend="nccl"`) over
xGMI.  The path shards over VIEWS: every rank holds a full replica of the anchors + MLPs,
rank r renders view (step * world + r), and a sum all-reduce of the gradient per step keeps
the replicas identical (444 B per anchor + 0.33 MB of MLP weights: 0.44 GB at 1 M anchors).
Few, large collectives on purpose: xGMI is point-to-point, a ring all-reduce is bound by one
~153 GB/s link per hop, and many small buckets would pay the per-collective latency once
each without adding bandwidth — the six per-anchor tensors go in place, the ~40 small MLP
tensors as one flat bucket.

No collective is needed inside a view (prefilter -> expand -> rasterize -> backward is
rank-local), and encode/decode shard over chunk streams (codec_driver).  Everything here
works with the gloo backend on CPU tensors too, which is how tests/test_dist.py covers it.
"""
from __future__ import annotations

import contextlib
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


_LOCAL_ONLY = 0


@contextlib.contextmanager
def local_only():
    """Inside this block the helpers of this module behave as in a single process (world 1, rank 0, no
    collectives) even though a process group exists: for work ONE rank does on its own, e.g. bench.py's codec leg
    on rank 0 while the other ranks wait at the next barrier."""
    global _LOCAL_ONLY
    _LOCAL_ONLY += 1
    try:
        yield
    finally:
        _LOCAL_ONLY -= 1


def world():
    return dist.get_world_size() if not _LOCAL_ONLY and dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if not _LOCAL_ONLY and dist.is_available() and dist.is_initialized() else 0


def view_for(step: int, n_views: int, r: int | None = None, w: int | None = None) -> int:
    """Index of the camera rank r renders at `step`: consecutive views go to consecutive
    ranks, so one step covers `world` distinct views (reference: one random view per
    iteration, train.py:152)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    return (step * w + r) % n_views


def shard(items: Sequence, r: int | None = None, w: int | None = None) -> list:
    """Round-robin shard of independent units (eval views, codec chunk streams)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    return [it for i, it in enumerate(items) if i % w == r]


BIG_TENSOR = 1 << 20        # elements; per-anchor tensors are far above, MLP weights far below


def allreduce_gradients(params: Iterable[torch.nn.Parameter], average: bool = True) -> int:
    """Sum (or average) .grad of every parameter across ranks.  The per-anchor tensors (tens to hundreds of MB
    each, 99.9 % of the bytes) are reduced IN PLACE, one collective each — large enough to run at link
    bandwidth, and nothing is copied into or out of a bucket; the many small MLP tensors share ONE flat bucket.
    Which path a parameter takes depends on its SIZE only, so every rank issues the same collectives in the same
    order.  Parameters whose .grad is None on this rank contribute zeros, so ranks may differ in which parameters
    received gradients (e.g. anchors invisible from one view).  Returns the number of elements reduced."""
    w = world()
    params = [p for p in params if p.requires_grad]
    if w == 1 or not params:
        return 0
    dev = params[0].device
    avg_in_collective = average and dist.get_backend() == "nccl"       # RCCL averages inside the collective
    op = dist.ReduceOp.AVG if avg_in_collective else dist.ReduceOp.SUM
    # a parameter NO rank produced a gradient for keeps .grad = None (the optimiser skips it, as on one GPU: mlp_grid /
    # latent_codec / _hyper_latent before iteration 10 000); zeros are filled in only where SOME rank has a gradient
    has = _any_rank_has_grad(params, dev)
    params = [p for p, h in zip(params, has) if h]
    pending = []
    for p in params:
        if p.numel() >= BIG_TENSOR:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            elif not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
            pending.append((p.grad, dist.all_reduce(p.grad, op=op, async_op=True)))
    small = [p for p in params if p.numel() < BIG_TENSOR]
    sizes = [p.numel() for p in small]
    if small:
        # no zero fill (only gradient-less parameters are zeroed) and the reduced bucket BECOMES the gradients (views)
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        off = 0
        for p, n in zip(small, sizes):
            if p.grad is not None:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            else:
                flat[off:off + n].zero_()
            off += n
        pending.append((flat, dist.all_reduce(flat, op=op, async_op=True)))
        off = 0
        for p, n in zip(small, sizes):
            p.grad = flat[off:off + n].view_as(p)
            off += n
    for t, work in pending:
        work.wait()
        if average and not avg_in_collective:
            t /= w
    return int(sum(p.numel() for p in params))


def _any_rank_has_grad(params, dev) -> list:
    """[some rank has .grad for p  for p in params]: ONE small MAX all-reduce (host tensor on gloo)."""
    local = [0 if p.grad is None else 1 for p in params]
    on_dev = dist.get_backend() == "nccl"
    m = torch.tensor(local, dtype=torch.int32, device=dev if on_dev else "cpu")
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return [bool(v) for v in m.tolist()]


_OPEN_SYNC = None          # weakref to the GradientSync that currently owns the hooks / the deferral switch

# The rows (anchors) this rank's view produced gradients for, noted by renderer.generate_neural_gaussians for the phases in
# which only visible anchors are differentiated (step <= 10 000: no context model over all anchors); None = every row.
# GradientSync(sparse="auto") reads it: the decision to take the touched-rows path is the PHASE (the same on every rank),
# whether the compact exchange pays is decided on the UNION of the ranks' rows (also the same on every rank).
_TOUCHED = {"rows": None, "n": 0}


def note_touched_rows(mask, n_rows: int = 0):
    """mask: bool [N] of the anchors whose per-anchor gradients this view can touch, or None for "all of them"."""
    _TOUCHED["rows"], _TOUCHED["n"] = mask, int(n_rows)


def _auto_rows(p):
    m = _TOUCHED["rows"]
    if m is None or p.dim() < 1 or p.shape[0] != m.shape[0]:
        return None
    return m


class GradientSync:
    """Gradient all-reduce that starts DURING the backward (SURVEY 8e "overlap with the tail of backward").

    Every parameter gets a post-accumulate-grad hook; a parameter whose gradient is final is reduced as soon as all
    parameters BEFORE it in a fixed issue order have been issued (head-of-line), so that every rank issues the same
    collectives in the same order whatever order its own autograd engine finishes them in — and even if one of them
    never receives a gradient on some rank (that rank contributes zeros from `finish()`).  The issue order starts
    as the parameter order and is replaced after the first step by the order in which rank 0 saw the gradients
    become final (broadcast once), which removes the head-of-line waits.  Large tensors (the per-anchor ones) go in
    place, one collective each; the small MLP / prior tensors are packed into one flat bucket in `finish()`.
    On RCCL the collectives run on the process group's own stream behind the producing kernels, i.e. beside the
    rest of the backward; what this buys on the ContextGS step is bounded — the six per-anchor gradients receive
    their last contribution from the context model's backward, the final nodes of the graph — and is measured by
    the driver's scaling run, not here.

    sparse: optional callable(param) -> bool row mask (or None) of the rows this rank touched; parameters for which
    every rank touched at most `sparse_below` of the rows are reduced over the UNION of touched rows only (one
    byte-mask all-reduce + a compact all-reduce), e.g. before iteration 10 000, when a view only produces gradients
    for the anchors it sees."""

    def __init__(self, params: Iterable[torch.nn.Parameter], average: bool = True, sparse="auto",
                 sparse_below: float = 0.5, defer_weight_gradients: bool = True, big_mode: str = "per_tensor"):
        self.params = [p for p in params if p.requires_grad]
        # big_mode: "per_tensor" (in place, one collective per per-anchor tensor, started from the gradient hooks: overlaps the
        # tail of the backward) or "bucket" (all of them packed into ONE flat buffer in finish(): one collective, two extra
        # passes over the payload, no overlap) — choose_big_mode() picks from measured collective latency / bandwidth
        assert big_mode in ("per_tensor", "bucket")
        self.big_mode = big_mode
        # sparse="auto" (default since round 5): the rows the renderer noted for this step (note_touched_rows) in the phases
        # whose gradients only reach visible anchors; a callable(param) -> row mask | None as before; None = always dense
        self.average, self.sparse, self.sparse_below = average, (_auto_rows if sparse == "auto" else sparse), sparse_below
        self._union = None                  # (row-mask object, union index | None) of this step: one mask exchange per step
        self._viol = []                     # (param, device bool: gradient outside the noted rows) of this step's compact exchanges
        self._dense_only = set()            # ids of parameters that showed gradient outside the noted rows once: dense from then on
        self.exposed_ms, self._exposure = [], None      # per step: how long the compute stream stood behind the collectives
        self.big = [p for p in self.params if p.numel() >= BIG_TENSOR] if big_mode == "per_tensor" else []
        self.small = [p for p in self.params if p.numel() < BIG_TENSOR] if big_mode == "per_tensor" else list(self.params)
        self.order = list(range(len(self.big)))              # issue order = indices into self.big
        self._pos = {id(p): k for k, p in enumerate(self.big)}
        self._ready, self._seen, self._next, self._pending = set(), [], 0, []
        self._order_synced = False
        # big tensors expected to receive a gradient on some rank this step = those that did last step (agreed on by all
        # ranks through finish()'s mask all-reduce); only these are issued from the hooks, so a tensor nobody differentiates
        # (e.g. _hyper_latent before iteration 10 000) costs no zero all-reduce and keeps .grad = None
        self._active = set(range(len(self.big)))
        self._filled = set()
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.big]
        self.bytes_reduced = 0
        # Nothing in the backward depends on the MLP weight gradients (≈ 1.1 ms of kernels per step at 1 M anchors), while
        # the per-anchor gradients become final in the graph's last nodes: with more than one rank the MLP nodes leave
        # their weight-gradient launches for the end of the backward (mlp.defer_weight_gradients), this object issues
        # whatever per-anchor collectives are still waiting first (_issue_leftovers), and the launches then run on the
        # compute stream BESIDE the collectives on RCCL's stream.  Gradients are bit-identical either way.
        #  the deferral is process-global state in mlp.py: it is scoped to THIS object — a sync that is replaced
        # (a new one after adjust_anchor) or garbage-collected without close() must not leave deferral on with
        # a hook that issues collectives for stale Parameters.  The hook only holds a weak reference, a new GradientSync closes
        # the one that is still open, and __del__ closes too.
        #  only a sync that is being REPLACED is closed for the caller: one whose parameters overlap the new one's
        # (the same model again) or are stale (storage swapped by optimizer surgery: no .grad hook can fire for them any more
        # through the model).  A second sync that is alive on purpose (another model, an evaluation replica) keeps its hooks —
        # but the weight-gradient deferral switch of mlp.py is one per process, so the two share it and a warning says so.
        global _OPEN_SYNC
        prev_sync = _OPEN_SYNC() if _OPEN_SYNC is not None else None
        if prev_sync is not None and prev_sync._handles:
            mine = {id(p) for p in self.params}
            if any(id(p) in mine for p in prev_sync.params) or not prev_sync.params:
                prev_sync.close()
            else:
                import warnings
                warnings.warn("contextgs_amd.dist.GradientSync: another GradientSync over different parameters is still open; "
                              "both stay active (close() the old one if it was meant to be replaced)", stacklevel=2)
        self._deferring = self._hook = None
        if defer_weight_gradients and world() > 1:
            import weakref
            from . import mlp
            self._deferring = mlp.defer_weight_gradients(True)
            ref = weakref.ref(self)

            def hook():
                s = ref()
                if s is not None:
                    s._issue_leftovers()
            self._hook = mlp.add_before_flush_hook(hook)
        import weakref as _wr
        _OPEN_SYNC = _wr.ref(self)

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []
        if self._deferring is not None:
            from . import mlp
            mlp.remove_before_flush_hook(self._hook)
            #  the deferral switch is one per process: a sync that is closed late (garbage collection of a replaced
            # one) must not turn it off under the sync that is open NOW
            cur = _OPEN_SYNC() if _OPEN_SYNC is not None else None
            if cur is None or cur is self or cur._deferring is None:
                mlp.defer_weight_gradients(self._deferring)
            self._deferring = self._hook = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- internals ------------------------------------------------------------------------------------------------
    def _op(self):
        in_coll = self.average and dist.get_backend() == "nccl"
        return (dist.ReduceOp.AVG if in_coll else dist.ReduceOp.SUM), in_coll

    def _issue(self, p):
        if p.grad is None:
            p.grad = torch.zeros_like(p)
            self._filled.add(id(p))
        elif not p.grad.is_contiguous():
            p.grad = p.grad.contiguous()
        op, in_coll = self._op()
        rows = None
        if self.sparse is not None and p.dim() >= 1 and p.shape[0] > 1 and id(p) not in self._dense_only:
            rows = self.sparse(p)
        if rows is not None:
            self._pending.append(("rows", p, *self._issue_rows(p, rows, op)))
        else:
            self._pending.append(("dense", p, p.grad, dist.all_reduce(p.grad, op=op, async_op=True)))
            self.bytes_reduced += p.grad.numel() * p.grad.element_size()

    def _issue_rows(self, p, rows, op):
        """Union of the ranks' touched rows (byte mask, MAX), then a compact all-reduce of those rows if it pays.  The union of
        one row-mask object is exchanged once per step and shared by every tensor that uses it (the six per-anchor tensors of
        a view share the visible-anchor mask)."""
        if self._union is not None and self._union[0] is rows:
            idx = self._union[1]
        else:
            m = rows.to(torch.uint8).contiguous()
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
            idx = torch.nonzero(m)[:, 0]
            self.bytes_reduced += m.numel()
            self._union = (rows, idx)
        if idx.numel() > self.sparse_below * p.shape[0]:
            self.bytes_reduced += p.grad.numel() * p.grad.element_size()
            return None, p.grad, dist.all_reduce(p.grad, op=op, async_op=True)
        #  the noted rows are what the RENDERER differentiates; a loss term outside it (train.py:209's mask
        # regulariser on `_mask`, 3000 < step <= 10000) puts gradient on every row.  Rows outside the union are not exchanged
        # here, so: note (on the device, no host read) whether this rank has any; finish() takes the MAX over ranks with the
        # has-gradient mask it exchanges anyway and repairs such a tensor with one dense collective, after which the tensor
        # stays on the dense path.
        outside = torch.ones(p.shape[0], dtype=torch.bool, device=p.grad.device)
        outside[idx] = False
        self._viol.append((p, (p.grad.reshape(p.shape[0], -1).ne(0).any(dim=1) & outside).any()))
        compact = p.grad.index_select(0, idx).contiguous()
        self.bytes_reduced += compact.numel() * compact.element_size()
        return idx, compact, dist.all_reduce(compact, op=op, async_op=True)

    def _drain_ready(self):
        while self._next < len(self.order):
            k = self.order[self._next]
            if k not in self._active:                      # issued (if at all) from finish(), after the mask
                self._next += 1
                continue
            if k not in self._ready:
                break
            self._issue(self.big[k])
            self._next += 1

    def _issue_leftovers(self):
        """Every predicted-active per-anchor tensor not issued by the hooks yet, in issue order (a rank without a gradient
        for one contributes zeros).  Runs at the end of the backward — from the deferred-weight-gradient flush, else from
        finish() — at the same point of every rank's collective sequence."""
        if world() == 1:
            return
        while self._next < len(self.order):
            k = self.order[self._next]
            if k in self._active:
                self._issue(self.big[k])
            self._next += 1

    def _on_grad(self, p):
        if world() == 1:
            return
        k = self._pos[id(p)]
        self._ready.add(k)
        self._seen.append(k)
        self._drain_ready()

    # -- API ----------------------------------------------------------------------------------------------------------
    def finish(self) -> int:
        """Call after loss.backward(): issues whatever is left (a parameter without a gradient on this rank contributes
        zeros IF some other rank has one; a parameter no rank differentiated keeps .grad = None, so Adam skips it exactly
        as on one GPU), reduces the small-tensor bucket, waits for everything.  Returns the bytes that went through
        collectives."""
        w = world()
        self.bytes_reduced = self.bytes_reduced if w > 1 else 0
        if w == 1:
            self._reset()
            return 0
        self._issue_leftovers()                               # gradient-less (on this rank) or out-of-order leftovers
        # which parameters have a gradient on SOME rank (one small MAX all-reduce, at the same point of every rank's
        # collective sequence: after the predicted-active big tensors)
        every = self.big + self.small
        local = [0 if (p.grad is None or id(p) in self._filled) else 1 for p in every]
        on_dev = dist.get_backend() == "nccl"
        m = torch.tensor(local, dtype=torch.int32, device=every[0].device if on_dev else "cpu")
        # + one entry per big tensor: "a compact exchange of this step left gradient outside the exchanged rows on some rank"
        viol_local = torch.zeros(len(self.big), dtype=torch.int32, device=m.device)
        for p_, flag in self._viol:
            viol_local[self._pos[id(p_)]] = flag.to(device=m.device, dtype=torch.int32)
        m = torch.cat([m, viol_local])
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        flags = m.tolist()
        has, viol = [bool(v) for v in flags[:len(every)]], [bool(v) for v in flags[len(every):]]
        has_big, has_small = has[:len(self.big)], has[len(self.big):]
        for k in range(len(self.big)):                       # mispredicted inactive: reduce now (same order on every rank)
            if has_big[k] and k not in self._active:
                self._issue(self.big[k])
        op, in_coll = self._op()
        flat = None
        small = [p for p, h in zip(self.small, has_small) if h]
        if small:
            sizes = [p.numel() for p in small]
            flat = torch.empty(sum(sizes), dtype=torch.float32, device=small[0].device)
            off = 0
            for p, n in zip(small, sizes):
                if p.grad is not None:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
                else:
                    flat[off:off + n].zero_()
                off += n
            work = dist.all_reduce(flat, op=op, async_op=True)
            self.bytes_reduced += flat.numel() * 4
            off = 0
            for p, n in zip(small, sizes):
                p.grad = flat[off:off + n].view_as(p)
                off += n
            self._pending.append(("dense", None, flat, work))
        ev0 = ev1 = None
        t_host = 0.0
        _first = self._pending[0] if self._pending else None
        if _first is not None and (_first[2] if _first[0] == "dense" else _first[3]).is_cuda:
            # how long the compute stream stands behind the collectives: an event before the first wait and one after the last
            # (on RCCL `wait()` only makes the current stream wait for the communicator's stream)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        else:
            import time as _time
            t_host = _time.perf_counter()
        for kind, p, *rest in self._pending:
            if kind == "dense":
                t, work = rest
                work.wait()
                if self.average and not in_coll:
                    t /= w
            else:
                idx, t, work = rest
                work.wait()
                if self.average and not in_coll:
                    t /= w
                if idx is not None:
                    p.grad.index_copy_(0, idx, t)
                    if viol[self._pos[id(p)]]:
                        # some rank has gradient outside the exchanged rows: reduce those rows now (the union's rows, already
                        # reduced, enter as zeros and are restored afterwards); same decision and order on every rank
                        self._dense_only.add(id(p))
                        rest = p.grad.clone()
                        rest[idx] = 0
                        dist.all_reduce(rest, op=op)
                        if self.average and not in_coll:
                            rest /= w
                        rest[idx] = t
                        p.grad.copy_(rest)
                        self.bytes_reduced += rest.numel() * rest.element_size()
        if ev0 is not None:
            ev1.record()
            if self._exposure is not None:                    # the previous step's pair is complete by now: no sync here
                try:
                    self.exposed_ms.append(self._exposure[0].elapsed_time(self._exposure[1]))
                except RuntimeError:
                    pass
            self._exposure = (ev0, ev1)
        elif self._pending:
            import time as _time
            self.exposed_ms.append((_time.perf_counter() - t_host) * 1e3)
        for k, p in enumerate(self.big):                      # predicted active, but no rank had a gradient: the zeros
            if k in self._active and not has_big[k]:           # that kept the collective sequence aligned are dropped
                p.grad = None
        active = {k for k in range(len(self.big)) if has_big[k]}
        # (round 6) the completion order is a property of the PHASE: before iteration 10 000 the per-anchor gradients come out
        # of the expansion's backward, afterwards feat / scaling / offsets / anchor leave the context model's last kernels while
        # `_mask` is final right after the rate node — with the order of the first step ever kept for good, `_mask` waited
        # head-of-line behind tensors that finish a millisecond later.  Whenever the set of tensors with a gradient changes
        # (same decision on every rank: it comes from the MAX-reduced mask), the order is adopted again from rank 0's
        # completion order of THIS step.
        if active != self._active:
            self._order_synced = False
        self._active = active
        if not self._order_synced:                            # adopt rank 0's completion order from now on
            seen = self._seen + [k for k in range(len(self.big)) if k not in self._seen]
            self.order = broadcast_object(seen, src=0)
            self._order_synced = True
        n = self.bytes_reduced
        self._reset()
        return n

    def _reset(self):
        self._ready, self._seen, self._next, self._pending, self.bytes_reduced = set(), [], 0, [], 0
        self._filled = set()
        self._union = None
        self._viol = []

    def exposure_report(self):
        """Mean / max of the per-step time the compute stream waited for the collectives in finish() (ms), over the steps
        recorded so far; the list is cleared."""
        if self._exposure is not None:
            try:
                torch.cuda.synchronize()
                self.exposed_ms.append(self._exposure[0].elapsed_time(self._exposure[1]))
            except RuntimeError:
                pass
            self._exposure = None
        v, self.exposed_ms = self.exposed_ms, []
        return {"steps": len(v), "mean_ms": (sum(v) / len(v) if v else None), "max_ms": (max(v) if v else None)}


def choose_big_mode(report: dict, n_big_tensors: int = 6, copy_GBps: float = 3000.0) -> dict:
    """Per-tensor in-place all-reduces or one flat bucket for the per-anchor gradients?  From diagnostics()' measurements on
    THIS group: per tensor = n collectives whose fixed cost is the small all-reduce's latency, but they start from the gradient
    hooks and run beside the backward; bucket = one collective + a pack and an unpack pass over the payload (HBM copies at
    ~copy_GBps) with nothing overlapped.  Returns {"mode", "est_per_tensor_ms", "est_bucket_ms", ...}: the payload's transfer time
    is the same in both and cancels; what differs is (n - 1) collective latencies against two copy passes."""
    ar = report.get("all_reduce", {}) if report else {}
    big, small = ar.get("per_anchor_payload"), ar.get("small_bucket")
    if not big or not small:
        return {"mode": "per_tensor", "why": "no measurement"}
    lat = float(small["median_ms"])
    copy_ms = 2.0 * float(big["bytes"]) / (copy_GBps * 1e9) * 1e3
    est_pt, est_b = float(big["median_ms"]) + (n_big_tensors - 1) * lat, float(big["median_ms"]) + copy_ms
    return {"mode": "per_tensor" if est_pt <= est_b else "bucket", "est_per_tensor_ms": round(est_pt, 3),
            "est_bucket_ms": round(est_b, 3), "small_allreduce_latency_ms": round(lat, 4), "pack_unpack_ms": round(copy_ms, 3)}


_shared_rng_counter = [0]


def shared_rand_like(t: torch.Tensor) -> torch.Tensor:
    """torch.rand_like(t) with the SAME values on every rank (rank 0 draws, the others receive): for random decisions
    that must not differ between replicas, e.g. the candidate sub-sampling of anchor growing
    (scene/gaussian_model.py:769).  The per-rank device generators cannot be relied on for that: ranks render
    different views and draw different amounts of visible-sized noise, so their streams have diverged."""
    u = torch.rand_like(t)
    if world() > 1:
        if dist.get_backend() != "nccl" and u.is_cuda:       # gloo: through the host
            h = u.cpu()
            dist.broadcast(h, src=0)
            u.copy_(h)
        else:
            dist.broadcast(u, src=0)
    return u


def allreduce_stats(tensors: List[torch.Tensor]) -> None:
    """Sum densification statistics in place across ranks (offset_gradient_accum,
    offset_denom, opacity_accum, anchor_demon; scene/gaussian_model.py:696-713) so that
    every replica takes the same grow/prune decisions."""
    if world() == 1 or not tensors:
        return
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
        off += n


def broadcast_parameters(params: Iterable[torch.nn.Parameter], src: int = 0) -> None:
    """Make every replica start from rank `src`'s parameters."""
    if world() == 1:
        return
    #  a write through `.data` / a collective does not bump the tensor's version counter, which caches keyed on it
    # (EntropyBottleneck.update's tables) would then survive: write through an alias that shares the counter, bump it, and
    # tell the caches explicitly
    with torch.no_grad():
        for p in params:
            t = p.detach()
            dist.broadcast(t, src=src)
            t.add_(0)                            # in-place no-op: bumps the shared version counter
    from . import entropy_bottleneck as _eb
    _eb.invalidate_tables()


def stream_blocks(edges, w: int | None = None) -> list[int]:
    """Split S consecutive streams (edges: S+1 non-decreasing element offsets) into `w` CONTIGUOUS blocks of about
    equal element count: returns w+1 stream indices, rank r codes streams [b[r], b[r+1]).  Contiguous blocks keep a
    rank's elements one slice of the flat arrays and make the concatenation of the ranks' byte streams in rank order
    the single-GPU file."""
    w = world() if w is None else w
    e = torch.as_tensor(edges, dtype=torch.int64).cpu()
    S = int(e.numel()) - 1
    if S <= 0:
        return [0] * (w + 1)
    total = int(e[-1] - e[0])
    targets = e[0] + (torch.arange(1, w, dtype=torch.int64) * total) // w
    cuts = torch.searchsorted(e, targets, right=False).clamp_(0, S).tolist()
    b = [0] + cuts + [S]
    for i in range(1, len(b)):                       # monotone even for degenerate (empty) streams
        b[i] = max(b[i], b[i - 1])
    return b


def all_gather_rows(local: torch.Tensor, counts: Sequence[int]) -> torch.Tensor:
    """Concatenate every rank's 1-D `local` (rank r holds counts[r] elements) in rank order on every rank: the
    exchange step of the sharded decoder (a level's decoded values are the next level's context on all ranks).
    One padded all_gather over RCCL; gloo has no device all_gather, so there the data takes the host path."""
    w = world()
    if w == 1:
        return local
    m = max(int(c) for c in counts)
    on_host = dist.get_backend() != "nccl" and local.is_cuda
    buf = torch.zeros(m, dtype=local.dtype, device="cpu" if on_host else local.device)
    buf[: local.numel()].copy_(local)
    out = [torch.empty_like(buf) for _ in range(w)]
    dist.all_gather(out, buf)
    return torch.cat([o[: int(c)] for o, c in zip(out, counts)]).to(local.device)


def gather_objects(obj, dst: int = 0):
    """Python objects of all ranks on rank dst (list in rank order), None elsewhere."""
    w, r = world(), rank()
    if w == 1:
        return [obj]
    out = [None] * w if r == dst else None
    dist.gather_object(obj, out, dst=dst)
    return out


def broadcast_object(obj, src: int = 0):
    if world() == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def gather_bytes(chunks: List[bytes], dst: int = 0) -> List[bytes] | None:
    """Gather round-robin sharded byte streams back into their global order on rank dst
    (the container is a concatenation + length table, so order is the only constraint)."""
    w, r = world(), rank()
    if w == 1:
        return list(chunks)
    out = [None] * w if r == dst else None
    dist.gather_object(list(chunks), out, dst=dst)
    if r != dst:
        return None
    n = sum(len(o) for o in out)
    merged = [None] * n
    for rr, lst in enumerate(out):
        for j, b in enumerate(lst):
            merged[j * w + rr] = b
    return merged


def diagnostics(payload_bytes: int, small_bytes: int = 350_000, reps: int = 5, device=None) -> dict | None:
    """What a multi-GPU run should print BEFORE its timed region so that its scaling numbers can be read:
    the ranks and devices the process group reports, and the measured time / bus bandwidth of one dense gradient all-reduce
    of `payload_bytes` (444 B per anchor) and of the small-tensor bucket.  Bus bandwidth = 2 (G - 1) / G x bytes / time (the
    ring's per-link traffic), to be compared with DESIGN.md section 5's per-link figures.  Collective on every rank; returns
    the report on every rank (None at world 1)."""
    import time as _time
    w = world()
    if w == 1:
        return None
    on_dev = dist.get_backend() == "nccl"
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if on_dev else torch.device("cpu"))
    name = torch.cuda.get_device_name(dev) if dev.type == "cuda" else "cpu"
    ranks = gather_objects({"rank": rank(), "device": str(dev), "name": name, "pid": __import__("os").getpid()}, dst=0)
    out = {"world": w, "backend": dist.get_backend(), "ranks": broadcast_object(ranks, src=0), "all_reduce": {}}
    for tag, nbytes in (("per_anchor_payload", int(payload_bytes)), ("small_bucket", int(small_bytes))):
        n = max(1, nbytes // 4)
        buf = torch.zeros(n, dtype=torch.float32, device=dev)
        dist.all_reduce(buf)                     # first use of this size
        times = []
        for _ in range(reps):
            if dev.type == "cuda":
                torch.cuda.synchronize()
            dist.barrier()
            t0 = _time.perf_counter()
            dist.all_reduce(buf)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            times.append(_time.perf_counter() - t0)
        t = torch.tensor([sorted(times)[len(times) // 2]], dtype=torch.float64, device=dev if on_dev else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sec = float(t.item())
        sig = lambda x: float("%.4g" % x)          # (four significant digits: a 64-byte probe on a slow host is ~1e-6 GB/s, not 0)
        out["all_reduce"][tag] = {"bytes": n * 4, "median_ms": sig(sec * 1e3),
                                  "algbw_GBps": sig(n * 4 / sec / 1e9),
                                  "busbw_GBps": sig(2.0 * (w - 1) / w * n * 4 / sec / 1e9)}
        del buf
    return out
