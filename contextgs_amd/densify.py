"""Densification statistics and anchor growing — SURVEY section 8(f) rank 1; drop-ins for
`GaussianModel.training_statis` (scene/gaussian_model.py:696-713) and `GaussianModel.anchor_growing`
(:762-855) of the reference, plus the optimizer surgery around them: `cat_tensors_to_optimizer` (:673-694),
`_prune_anchor_optimizer` / `prune_anchor` (:715-760) and `adjust_anchor` (:856-910), so that
`contextgs_amd.model.GaussianModel` densifies on its own (round 3).

* training_statis: one HIP pass over the visible slots (`cgs_densify_stats`, csrc/densify.hip) instead of ~15
  launches with three boolean-mask index_puts over all N*K offsets.
* prune / adjust_anchor: the eight per-anchor parameters, their Adam moments and the four statistics buffers are
  compacted by ONE `cgs_compact_rows` launch over one `nonzero` of the keep mask (the reference boolean-indexes each of
  the ~28 tensors separately: a mask scan + gather pair per tensor).
* anchor_growing: same candidate selection, voxel rounding, per-voxel feature max and new-anchor attributes; the
  de-duplication of candidate voxels against the existing anchors — an O(candidates x N) all-pairs compare in 4096-row
  chunks in the reference (:793-802) — is a sorted-key membership test (`voxels_already_present`).  The absent
  `torch_scatter.scatter_max` is `scatter_reduce(amax)`.
Device tensors only.
"""
from __future__ import annotations

import math

import torch

from . import _lib
from . import dist as _dist
from .encodings import Quantize_anchor


@torch.no_grad()
def training_statis(pc, viewspace_point_tensor, opacity, update_filter, offset_selection_mask, anchor_visible_mask):
    K = int(pc.n_offsets)
    _lib.require_device(opacity, pc.opacity_accum)
    vis_idx = torch.nonzero(anchor_visible_mask)[:, 0]
    n_vis = int(vis_idx.shape[0])
    op = opacity.detach().reshape(-1)
    op = op if (op.dtype == torch.float32 and op.is_contiguous()) else op.float().contiguous()
    if op.numel() != n_vis * K:
        raise ValueError("training_statis: opacity must hold n_visible * n_offsets values")
    sel = offset_selection_mask.reshape(-1).to(torch.uint8).contiguous()
    sel_pos = (torch.cumsum(sel, 0, dtype=torch.int64) - sel).contiguous()
    uf = update_filter.reshape(-1).to(torch.uint8).contiguous()
    grad = viewspace_point_tensor.grad
    grad = grad if (grad.dtype == torch.float32 and grad.is_contiguous()) else grad.float().contiguous()
    if grad.dim() != 2 or grad.shape[1] != 3 or grad.shape[0] != uf.shape[0]:
        raise ValueError("training_statis: viewspace_point_tensor.grad must be [P,3] with P = len(update_filter)")
    for t in (pc.opacity_accum, pc.anchor_demon, pc.offset_gradient_accum, pc.offset_denom):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("training_statis: accumulators must be contiguous fp32")
    _lib.check(_lib.lib().cgs_densify_stats(n_vis, K, _lib.ptr(vis_idx), _lib.ptr(op), _lib.ptr(sel), _lib.ptr(sel_pos),
                                            _lib.ptr(uf), _lib.ptr(grad), _lib.ptr(pc.opacity_accum), _lib.ptr(pc.anchor_demon),
                                            _lib.ptr(pc.offset_gradient_accum), _lib.ptr(pc.offset_denom),
                                            _lib.current_stream()), "cgs_densify_stats")


def _voxel_keys(v: torch.Tensor) -> torch.Tensor:
    """int32 [n,3] voxel coordinates -> one int64 key each (21 bits per axis)."""
    v = v.to(torch.int64) + (1 << 20)
    return (v[:, 0] << 42) | (v[:, 1] << 21) | v[:, 2]


def voxels_already_present(candidates: torch.Tensor, grid_coords: torch.Tensor) -> torch.Tensor:
    """bool[M]: candidate voxel m equals some row of grid_coords ([N,3] int) — :793-802 without the M x N compare."""
    if candidates.numel() == 0:
        return torch.zeros(0, dtype=torch.bool, device=candidates.device)
    lim = 1 << 20
    both = max(int(candidates.abs().max()), int(grid_coords.abs().max()) if grid_coords.numel() else 0)
    if both >= lim:
        raise ValueError("voxel coordinates exceed the 21-bit key range")
    return torch.isin(_voxel_keys(candidates), _voxel_keys(grid_coords))


def _group_max(src: torch.Tensor, inverse: torch.Tensor, n_groups: int) -> torch.Tensor:
    """torch_scatter.scatter_max(src, inverse[:,None].expand_as(src), dim=0)[0]"""
    out = torch.zeros(n_groups, src.shape[1], dtype=src.dtype, device=src.device)
    return out.scatter_reduce_(0, inverse.unsqueeze(1).expand(-1, src.shape[1]), src, reduce="amax", include_self=False)


def _grow_round(i, anchor_q, offset, scale3, feat, hyper, cand_mask, voxel_size, K, init_factor, hier):
    """One depth of anchor_growing given the CURRENT model tensors; returns the new-anchor dict or None."""
    dev = anchor_q.device
    all_xyz = anchor_q.unsqueeze(1) + offset * scale3.unsqueeze(1)
    size_factor = init_factor // (hier ** i)
    cur_size = voxel_size * size_factor
    grid_coords = torch.round(anchor_q / cur_size).int()
    selected_xyz = all_xyz.view(-1, 3)[cand_mask]
    selected_grid = torch.round(selected_xyz / cur_size).int()
    uniq, inverse = torch.unique(selected_grid, return_inverse=True, dim=0)
    keep = ~voxels_already_present(uniq, grid_coords)
    cand_anchor = uniq[keep] * cur_size
    M = int(cand_anchor.shape[0])
    if M == 0:
        return None
    new_scaling = torch.log(torch.ones(M, 6, dtype=torch.float32, device=dev) * cur_size)
    new_rotation = torch.zeros(M, 4, dtype=torch.float32, device=dev)
    new_rotation[:, 0] = 1.0
    new_opacities = torch.full((M, 1), math.log(0.1 / 0.9), dtype=torch.float32, device=dev)     # inverse_sigmoid(0.1)
    rep = lambda t: t.unsqueeze(1).expand(-1, K, -1).reshape(-1, t.shape[1])[cand_mask]
    new_feat = _group_max(rep(feat), inverse, uniq.shape[0])[keep]
    new_hyper = _group_max(rep(hyper), inverse, uniq.shape[0])[keep]
    return {"anchor": cand_anchor.float(), "scaling": new_scaling, "rotation": new_rotation, "anchor_feat": new_feat,
            "hyper_latent": new_hyper, "offset": torch.zeros(M, K, 3, dtype=torch.float32, device=dev),
            "mask": torch.ones(M, K, 1, dtype=torch.float32, device=dev), "opacity": new_opacities, "depth": i}


def _candidates(i, grads, threshold, offset_mask, hier, rand):
    cand = (grads >= threshold * ((hier // 2) ** i)) & offset_mask
    return cand & (rand > (0.5 ** (i + 1)))


@torch.no_grad()
def growing_rounds(anchor, offset, scaling, feat, hyper, x_bound_min, x_bound_max, grads, threshold, offset_mask, voxel_size,
                   K, update_depth=3, init_factor=100, hier=4, rand_fn=None):
    """Functional form of anchor_growing (:762-855): the per-round new-anchor dicts, the model tensors being
    extended between rounds exactly as cat_tensors_to_optimizer would."""
    _lib.require_device(anchor, grads)
    rand_fn = rand_fn or (lambda i, like: _dist.shared_rand_like(like))     # the same draw on every replica
    init_length = anchor.shape[0] * K
    rounds = []
    for i in range(update_depth):
        cand = _candidates(i, grads, threshold, offset_mask, hier, rand_fn(i, grads))
        length_inc = anchor.shape[0] * K - init_length
        if length_inc == 0:
            if i > 0:
                continue                                    # :774-777 (rounds i > 0 only run once something was added)
        else:
            cand = torch.cat([cand, torch.zeros(length_inc, dtype=torch.bool, device=cand.device)])
        anchor_q = Quantize_anchor.apply(anchor, x_bound_min, x_bound_max)[0]
        d = _grow_round(i, anchor_q, offset, torch.exp(scaling[:, :3]), feat, hyper, cand, voxel_size, K, init_factor, hier)
        if d is None:
            continue
        rounds.append(d)
        anchor = torch.cat([anchor, d["anchor"]])
        offset = torch.cat([offset, d["offset"]])
        scaling = torch.cat([scaling, d["scaling"]])
        feat = torch.cat([feat, d["anchor_feat"]])
        hyper = torch.cat([hyper, d["hyper_latent"]])
    return rounds


@torch.no_grad()
def anchor_growing(pc, grads, threshold, offset_mask, rand_fn=None):
    """Drop-in for GaussianModel.anchor_growing (:762-855): mutates `pc` through its own cat_tensors_to_optimizer.
    rand_fn(i, like): the uniform draw of round i (:769); default = one draw shared by all replicas."""
    K = int(pc.n_offsets)
    init_length = pc.get_anchor.shape[0] * K
    rand_fn = rand_fn or (lambda i, like: _dist.shared_rand_like(like))
    for i in range(pc.update_depth):
        cand = _candidates(i, grads, threshold, offset_mask, pc.update_hierachy_factor, rand_fn(i, grads))
        length_inc = pc.get_anchor.shape[0] * K - init_length
        if length_inc == 0:
            if i > 0:
                continue
        else:
            cand = torch.cat([cand, torch.zeros(length_inc, dtype=torch.bool, device=cand.device)])
        d = _grow_round(i, pc.get_anchor, pc._offset, pc.get_scaling[:, :3], pc._anchor_feat, pc._hyper_latent, cand,
                        pc.voxel_size, K, pc.update_init_factor, pc.update_hierachy_factor)
        if d is None:
            continue
        d.pop("depth")
        M = d["anchor"].shape[0]
        z = torch.zeros(M, 1, dtype=torch.float32, device=grads.device)
        pc.anchor_demon = torch.cat([pc.anchor_demon, z], dim=0)
        pc.opacity_accum = torch.cat([pc.opacity_accum, z.clone()], dim=0)
        t = pc.cat_tensors_to_optimizer(d)
        pc._anchor, pc._scaling, pc._rotation = t["anchor"], t["scaling"], t["rotation"]
        pc._anchor_feat, pc._hyper_latent, pc._offset = t["anchor_feat"], t["hyper_latent"], t["offset"]
        pc._mask, pc._opacity = t["mask"], t["opacity"]


# ---- optimizer surgery (scene/gaussian_model.py:656-760) ---------------------------------------------------------------
_NOT_PER_ANCHOR = ("mlp", "conv", "feat_base", "encoding", "codec")          # :676, :718
_ATTR = {"anchor": "_anchor", "offset": "_offset", "mask": "_mask", "anchor_feat": "_anchor_feat",
         "hyper_latent": "_hyper_latent", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}


def _per_anchor_groups(optimizer):
    return [g for g in optimizer.param_groups if not any(s in g["name"] for s in _NOT_PER_ANCHOR)]


def compact_rows(tensors, idx, clamp_col0=None, clamp_max=0.0):
    """[t[idx] for t in tensors] (rows) for fp32 device tensors in ONE launch (`cgs_compact_rows`); clamp_col0[i] >= 0 caps
    columns >= clamp_col0[i] of tensor i at clamp_max on the way."""
    import ctypes as C
    if not tensors:
        return []
    n_keep = int(idx.shape[0])
    srcs = [t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().float().contiguous() for t in tensors]
    _lib.require_device(idx, *srcs)
    outs = [torch.empty((n_keep,) + tuple(t.shape[1:]), dtype=torch.float32, device=t.device) for t in srcs]
    widths = [max(1, int(t[0].numel())) if t.shape[0] else max(1, int(torch.Size(t.shape[1:]).numel())) for t in srcs]
    for lo in range(0, len(srcs), 32):
        hi = min(len(srcs), lo + 32)
        k = hi - lo
        sp = (C.c_void_p * k)(*[t.data_ptr() for t in srcs[lo:hi]])
        dp = (C.c_void_p * k)(*[t.data_ptr() for t in outs[lo:hi]])
        wd = (C.c_int * k)(*widths[lo:hi])
        cl = (C.c_int * k)(*[(-1 if clamp_col0 is None else int(clamp_col0[i])) for i in range(lo, hi)])
        _lib.check(_lib.lib().cgs_compact_rows(k, sp, dp, wd, cl, float(clamp_max), _lib.ptr(idx), n_keep,
                                               _lib.current_stream()), "cgs_compact_rows")
    return outs


@torch.no_grad()
def cat_tensors_to_optimizer(pc, tensors_dict):
    """:673-694 — append rows to every per-anchor parameter; Adam moments (where a state exists) get zero rows."""
    out = {}
    for group in _per_anchor_groups(pc.optimizer):
        assert len(group["params"]) == 1
        old = group["params"][0]
        ext = tensors_dict[group["name"]]
        state = pc.optimizer.state.get(old, None)
        new = torch.nn.Parameter(torch.cat((old.detach(), ext.to(old.dtype)), dim=0).requires_grad_(True))
        if state is not None and len(state) > 0:
            state["exp_avg"] = torch.cat((state["exp_avg"], torch.zeros_like(ext, dtype=old.dtype)), dim=0)
            state["exp_avg_sq"] = torch.cat((state["exp_avg_sq"], torch.zeros_like(ext, dtype=old.dtype)), dim=0)
            del pc.optimizer.state[old]
            pc.optimizer.state[new] = state
        elif old in pc.optimizer.state:
            del pc.optimizer.state[old]
        group["params"][0] = new
        out[group["name"]] = new
    return out


@torch.no_grad()
def replace_tensor_to_optimizer(pc, tensor, name):
    """:656-670 — swap one parameter for `tensor`, zeroing its Adam moments."""
    out = {}
    for group in pc.optimizer.param_groups:
        if group["name"] == name:
            old = group["params"][0]
            state = pc.optimizer.state.get(old, None)
            new = torch.nn.Parameter(tensor.requires_grad_(True))
            if state is not None:
                state["exp_avg"], state["exp_avg_sq"] = torch.zeros_like(tensor), torch.zeros_like(tensor)
                del pc.optimizer.state[old]
                pc.optimizer.state[new] = state
            group["params"][0] = new
            out[name] = new
    return out


@torch.no_grad()
def _prune_rows(pc, keep_idx, extra=()):
    """Keep rows keep_idx of every per-anchor parameter, of its Adam moments and of the `extra` row tensors: one launch.
    Returns ({group name: new Parameter}, [compacted extras])."""
    groups = _per_anchor_groups(pc.optimizer)
    tensors, clamps, slots = [], [], []
    for g in groups:
        p = g["params"][0]
        state = pc.optimizer.state.get(p, None)
        has = state is not None and "exp_avg" in state
        slots.append((g, p, state if has else None, len(tensors)))
        tensors.append(p)
        clamps.append(3 if g["name"] == "scaling" else -1)          # :741-745: scales[:, 3:] capped at 0.05 (log space, as is)
        if has:
            tensors += [state["exp_avg"], state["exp_avg_sq"]]
            clamps += [-1, -1]
    n_model = len(tensors)
    tensors += list(extra)
    clamps += [-1] * len(extra)
    outs = compact_rows(tensors, keep_idx, clamps, 0.05)
    new_params = {}
    for g, p, state, at in slots:
        new = torch.nn.Parameter(outs[at].requires_grad_(True))
        if state is not None:
            state["exp_avg"], state["exp_avg_sq"] = outs[at + 1], outs[at + 2]
            del pc.optimizer.state[p]
            pc.optimizer.state[new] = state
        elif p in pc.optimizer.state:
            del pc.optimizer.state[p]
        g["params"][0] = new
        new_params[g["name"]] = new
    return new_params, outs[n_model:]


def _assign(pc, params):
    for name, attr in _ATTR.items():
        if name in params:
            setattr(pc, attr, params[name])
    pc._level_cache = None
    pc._anchor_q_cache = None


@torch.no_grad()
def prune_anchor(pc, mask):
    """:747-760 — drop the anchors where mask is True (parameters + optimizer state)."""
    keep = torch.nonzero(~mask)[:, 0]
    params, _ = _prune_rows(pc, keep)
    _assign(pc, params)


_STATS = ("offset_gradient_accum", "offset_denom", "opacity_accum", "anchor_demon")


@torch.no_grad()
def reduce_statistics(pc):
    """Multi-GPU: make the four densification buffers hold the GLOBAL statistics on every rank — the sum over
    ranks of what each rank accumulated SINCE THE LAST CALL, on top of the carry-over the ranks already share.

    A plain in-place all-reduce is only right once: entries that stay below adjust_anchor's reset thresholds
    (offset_denom <= 40, anchor_demon <= 80) carry their — by then global — counts into the next round, and a second SUM
    would count that carry once per rank.  So the model remembers the buffers as they were when the replicas
    last agreed (`_stats_base`: set here and at the end of adjust_anchor, after its resets / compaction / padding) and only
    the difference to it is reduced.  Idempotent; a no-op without a process group."""
    if _dist.world() == 1:
        return
    bufs = [getattr(pc, n) for n in _STATS]
    base = getattr(pc, "_stats_base", None)
    if base is None or any(b.shape != t.shape for b, t in zip(base, bufs)):
        base = [torch.zeros_like(t) for t in bufs]
    deltas = [t - b for t, b in zip(bufs, base)]
    _dist.allreduce_stats(deltas)
    for t, b, d in zip(bufs, base, deltas):
        t.copy_(b + d)
    pc._stats_base = [t.clone() for t in bufs]


@torch.no_grad()
def adjust_anchor(pc, check_interval=100, success_threshold=0.8, grad_threshold=0.0002, min_opacity=0.005, rand_fn=None,
                  reduce_stats=True):
    """:856-910.  With a process group the four statistics buffers are summed over the ranks first (reduce_statistics), so
    that every replica grows and prunes the same anchors (the growing draw is shared, `dist.shared_rand_like`);
    reduce_stats=False when the caller already called reduce_statistics(pc)."""
    K = int(pc.n_offsets)
    if reduce_stats:
        reduce_statistics(pc)
    # ---- adding anchors (:858-876) ----
    grads = pc.offset_gradient_accum / pc.offset_denom
    grads[grads.isnan()] = 0.0
    grads_norm = torch.norm(grads, dim=-1)
    offset_mask = (pc.offset_denom > check_interval * success_threshold * 0.5).squeeze(dim=1)
    anchor_growing(pc, grads_norm, grad_threshold, offset_mask, rand_fn)
    n_new = int(pc.get_anchor.shape[0])
    dev = pc.offset_denom.device
    pc.offset_denom[offset_mask] = 0
    pc.offset_gradient_accum[offset_mask] = 0
    pad = n_new * K - int(pc.offset_denom.shape[0])
    if pad:
        z = torch.zeros(pad, 1, dtype=torch.float32, device=dev)
        pc.offset_denom = torch.cat([pc.offset_denom, z], dim=0)
        pc.offset_gradient_accum = torch.cat([pc.offset_gradient_accum, z.clone()], dim=0)
    # ---- pruning (:878-908): anchors seen often enough whose accumulated opacity stayed below the floor ----
    prune_mask = (pc.opacity_accum < min_opacity * pc.anchor_demon).squeeze(dim=1)
    anchors_mask = (pc.anchor_demon > check_interval * success_threshold).squeeze(dim=1)
    prune_mask = torch.logical_and(prune_mask, anchors_mask)
    pc.opacity_accum[anchors_mask] = 0.0                     # :894-896 (before the compaction, as in the reference)
    pc.anchor_demon[anchors_mask] = 0.0
    if prune_mask.shape[0] > 0:
        keep = torch.nonzero(~prune_mask)[:, 0]
        params, (od, oga, oa, ad) = _prune_rows(
            pc, keep, extra=(pc.offset_denom.view(-1, K), pc.offset_gradient_accum.view(-1, K), pc.opacity_accum, pc.anchor_demon))
        _assign(pc, params)
        pc.offset_denom, pc.offset_gradient_accum = od.view(-1, 1), oga.view(-1, 1)
        pc.opacity_accum, pc.anchor_demon = oa, ad
    pc.max_radii2D = torch.zeros(pc.get_anchor.shape[0], device=dev)
    if _dist.world() > 1:      # what the replicas agree on from here; the next reduce_statistics sums the ranks' additions to it
        pc._stats_base = [getattr(pc, n).clone() for n in _STATS]
