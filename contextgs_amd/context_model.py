"""Anchor-level hierarchical context model: drop-in for
the module-level functions of the reference's scene/gaussian_model.py:1541-1792
(`multi_scale_generating`, `extract_context_feat`, `find_divide_scale`,
`divide_levels`, `mapping_to_orign`, `index_of_level_L_in_orign`) and
utils/multi_level.py (`torch_unique_with_indices`).

Same level semantics including the reference's quirks (level-1
context rows are paired in lexicographic-voxel order vs ascending original
index; Q4: the training variant zeroes masked anchors before level 1; Q5:
fp32 left-to-right voxel key).  Quantisation and rate run in the fused HIP
kernels of elementwise.hip (encodings.STE_multistep, entropy_models), the grid
MLPs in the fused fp32-MFMA kernels of mlp.hip.

Data layout of the fast path: the rows each level codes are disjoint and
together cover every anchor, so the per-anchor tensors are gathered ONCE into
"coding order" (level L-1 rows, then L-2, ..., then level 0) and every level
works on a contiguous slice; context rows are gathered from the already-coded
prefix; the N-ordered outputs the reference returns are one inverse gather at
the end (or none: the renderer composes it with its visibility gather).  The
reference instead scatters into / gathers from N-sized buffers at every level
(:1569-1647), ~30 full-size passes per call.

Cited lines are scene/gaussian_model.py unless stated otherwise.
"""
from __future__ import annotations

import torch

from . import _lib
from . import ctx_ops as _ctx
from . import encodings as _enc
from . import mlp as _mlp
from .encodings import STE_multistep, get_binary_vxl_size
from .entropy_bottleneck import EntropyBottleneck as _EntropyBottleneck
from .multi_level import torch_unique_with_indices

Q_FEAT0, Q_SCALING0, Q_OFFSETS0 = 1, 0.001, 0.2      # :1564-1566
import os as _os
LAZY_MODE = int(_os.environ.get("CGS_LAZY_MODE", "2"))   # tuning knob: 0 gather all, 1 defer feat, 2 defer feat+scaling+offsets
FUSED_TRAINING = True      # fused HIP stages for the training path (tests flip it to compare with the torch composition)
HYPER_BLOCKS = _os.environ.get("CGS_HYPER_BLOCKS", "1") != "0"   # the noisy hyper latents leave their node one block per level
ROW_SOURCE = _os.environ.get("CGS_ROW_SOURCE", "1") != "0"      # A/B knob: 0 = gather into coding order first
RATE_SIDE = _os.environ.get("CGS_RATE_SIDE", "1") != "0"        # A/B knob: 0 = rate gradients through autograd (dense buffers + adds)
LEVEL_FUSED = _os.environ.get("CGS_LEVEL_FUSED", "1") != "0"    # A/B knob: 0 = round 4's rowcat -> mlp2 -> noise_quant launches per level
RATE_FUSED = _os.environ.get("CGS_RATE_FUSED", "1") != "0"      # A/B knob: 0 = round 5's gather -> mlp2 -> level_rate launches on the rate subset
EARLY_LEVELS = _os.environ.get("CGS_EARLY_LEVELS", "1") != "0"  # A/B knob: 0 = the level kernels are enqueued behind the rate subset's read-back (rounds 5-6a)


def _compose_down(index, maps):
    """index (into level len(maps)) -> original-space index: index -> maps[-1][index] -> maps[-2][.] -> ... -> maps[0][.]"""
    for m in reversed(maps):
        index = m[index]
    return index


def mapping_to_orign(mapping_list, L, mask=None):                       # :1768-1787
    """Original-space indices of the anchors of level L (of those selected by `mask`).  mapping_list[k] holds, for
    every anchor of level k+1, its representative's index in level k."""
    if L <= 0:
        raise AssertionError("If L=0, the orgin space can be directly obtained")
    top = mapping_list[L - 1]
    return _compose_down(top if mask is None else top[mask], mapping_list[:L - 1])


def index_of_level_L_in_orign(mapping_list, inverse_indices_list, to_be_gathered_index, L):   # :1789-1792
    """Original-space index of the level-L ancestor of each original anchor: L hops up through the voxel
    membership tables, then L hops back down through the representatives."""
    up = to_be_gathered_index
    for inv in inverse_indices_list[:L]:
        up = inv[up]
    return _compose_down(up, mapping_list[:L])


def _voxel_keys(anchor, voxel_size, scale):
    """Level key of every anchor (:1732,1760; quirk Q5): fp32, two divisions left to right, round half to even.
    `scale` is a 0-d tensor in the scale search and a Python float afterwards, exactly as in the reference — the two
    are NOT the same fp32 operation on the device (a host scalar divides as a reciprocal multiplication)."""
    return torch.round(anchor / voxel_size / scale)


def find_divide_scale(pc, anchor, target_ratio, level_num):             # :1726-1749
    """One voxel-scale multiplier per coarser level: bisection on the scale until the level keeps
    target_ratio +- 0.01 of the previous level's anchors (or the bracket is narrower than 1).  The bracket of level
    k+1 starts at level k's result; levels are nested (level k+1 is searched on level k's voxel centres)."""
    hi0 = ((pc.x_bound_max - pc.x_bound_min) / pc.voxel_size).max()        # 0-d tensor: the bracket stays on the device
    scales, lo, pts = [], 1, anchor
    for _level in range(level_num - 1):
        hi = hi0
        while True:
            mid = (hi + lo) / 2
            cells = torch_unique_with_indices(_voxel_keys(pts, pc.voxel_size, mid), dim=0)[0]
            kept = cells.shape[0] / pts.shape[0]
            if abs(kept - target_ratio) < 0.01 or (hi - lo).abs() < 1:
                break
            if kept < target_ratio:
                hi = mid          # too few cells: voxels too large
            else:
                lo = mid
        pts = cells * pc.voxel_size * mid                                  # next level is searched on these centres
        lo = mid
        scales.append(mid.item())
    return scales


def divide_levels(pc, anchor, mask_anchor_bool=None):                   # :1751-1765
    """Per level k >= 1: the voxel representatives (`mapping`: smallest original index of each occupied voxel, in
    lexicographic voxel order) and every level-(k-1) anchor's voxel (`inverse`).  Training variant (mask given, quirk
    Q4): masked anchors are moved to the origin before level 1 instead of being dropped."""
    level_anchor = anchor
    anchors, inverses, mappings = [anchor], [], []
    for k in range(1, pc.level_num):
        if k == 1 and mask_anchor_bool is not None:
            level_anchor = level_anchor * mask_anchor_bool.unsqueeze(1)
        _cells, inverse, first, _counts = torch_unique_with_indices(
            _voxel_keys(level_anchor, pc.voxel_size, pc.level_scale[k - 1]), dim=0)
        level_anchor = level_anchor[first]
        anchors.append(level_anchor)
        inverses.append(inverse)
        mappings.append(first)
    return anchors, inverses, mappings, level_anchor


def _context_index(n, already_coded, inverse_indices_list, mapping_list, i):
    """Original-space indices of the context rows extract_context_feat(i) gathers (:1713-1721)."""
    dev = already_coded.device
    if i > 1:
        to_be_gathered_mask = torch.zeros(n, dtype=torch.bool, device=dev)
        to_be_gathered_mask.index_fill_(0, mapping_to_orign(mapping_list, i - 1), True)    # (no blocking scalar upload)
    else:
        to_be_gathered_mask = torch.ones(n, dtype=torch.bool, device=dev)
    to_be_gathered_mask = to_be_gathered_mask & (~already_coded)
    return index_of_level_L_in_orign(mapping_list, inverse_indices_list, torch.nonzero(to_be_gathered_mask)[:, 0], i)


def extract_context_feat(anchor_after_Q, feat_after_Q, grid_scaling_after_Q, already_coded, inverse_indices_list,
                         mapping_list, i):                              # :1711-1724
    idx = _context_index(anchor_after_Q.shape[0], already_coded, inverse_indices_list, mapping_list, i)
    # gather three row blocks instead of materialising the [N,59] concat (:1712) every level
    return torch.cat([anchor_after_Q[idx], feat_after_Q[idx], grid_scaling_after_Q[idx]], dim=1)


def context_rows(pc, anchor_after_Q, feat_after_Q, grid_scaling_after_Q, already_coded, inverse_indices_list, mapping_list, i):
    """extract_context_feat with the row index taken from the level plan's cache (level_plan was just called on these anchors:
    `ctx_idx[i]` is exactly _context_index at this point of the level loop) — no boolean mask, no torch.nonzero and therefore
    no host read per level (the container's level loop otherwise drains the coder launch it has just queued)."""
    cache = getattr(pc, "_level_cache", None)
    idx = None
    if cache is not None and tuple(cache["anchor"].shape) == tuple(anchor_after_Q.shape):
        idx = cache["ctx_idx"].get(i)
    if idx is None:
        return extract_context_feat(anchor_after_Q, feat_after_Q, grid_scaling_after_Q, already_coded, inverse_indices_list,
                                    mapping_list, i)
    return torch.cat([anchor_after_Q[idx], feat_after_Q[idx], grid_scaling_after_Q[idx]], dim=1)


def level_plan(pc, anchor, mask_anchor_bool):
    """Index bookkeeping of the level loop (:1559-1593), shared by the rate model, the
    encoder and the decoder.  Returns per level (from L-1 down to 0) the original-space
    indices to code.

    The division depends only on (anchor, mask, voxel_size, level_scale); anchors are frozen
    after densification (position lr = 0, arguments/__init__.py:86-87) and the anchor mask
    changes rarely, so the plan of the previous call is reused when both tensors compare equal
    (two elementwise compares + one host read instead of two device sorts and ~40 index
    kernels; SURVEY §8a "caching opportunity")."""
    c = _cached_plan(pc, anchor, mask_anchor_bool)
    return c["plan"], c["inverse"], c["mapping"]


def _cached_plan(pc, anchor, mask_anchor_bool):
    key = (float(pc.voxel_size), tuple(float(v) for v in pc.level_scale), int(pc.level_num),
           tuple(anchor.shape), mask_anchor_bool is None)
    cache = getattr(pc, "_level_cache", None)
    if cache is not None and cache["key"] == key:
        same = (anchor == cache["anchor"]).all()
        if mask_anchor_bool is not None:
            same = same & (mask_anchor_bool == cache["mask"]).all()
        if bool(same):
            return cache
    plan, inverse_indices_list, mapping_list = _level_plan_uncached(pc, anchor, mask_anchor_bool)
    n = anchor.shape[0]
    dev = anchor.device
    # coding order: perm = [rows of level L-1, rows of level L-2, ..., rows of level 0]
    origs = [p[2] for p in plan]
    perm = torch.cat(origs) if origs else torch.zeros(0, dtype=torch.long, device=dev)
    inv_perm = torch.zeros(n, dtype=torch.long, device=dev)
    inv_perm[perm] = torch.arange(perm.shape[0], device=dev)
    sizes = [int(o.shape[0]) for o in origs]
    # context rows (original space, and as positions in coding order) that the level coded AFTER level i needs
    already = torch.zeros(n, dtype=torch.bool, device=dev)
    ctx_idx, ctx_pos, ctx_csr = {}, {}, {}
    coded = 0
    for (i, _tc, orig, _a) in plan:
        already.index_fill_(0, orig, True)
        coded += int(orig.shape[0])
        if i != 0:
            idx = _context_index(n, already, inverse_indices_list, mapping_list, i)
            ctx_idx[i] = idx
            ctx_pos[i] = inv_perm[idx]
            # children of every coded row (CSR), for the atomics-free backward of the context gather
            order = torch.argsort(ctx_pos[i], stable=True)
            counts = torch.bincount(ctx_pos[i], minlength=coded)
            offs = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), counts.cumsum(0)])
            ctx_csr[i] = (offs, order, perm[:coded].contiguous())
    cache = dict(key=key, anchor=anchor.detach().clone(),
                 mask=None if mask_anchor_bool is None else mask_anchor_bool.clone(), plan=plan,
                 inverse=inverse_indices_list, mapping=mapping_list, perm=perm, inv_perm=inv_perm, sizes=sizes,
                 ctx_idx=ctx_idx, ctx_pos=ctx_pos, ctx_csr=ctx_csr, covers_all=bool(perm.shape[0] == n),
                 # anchors already stored in coding order (e.g. a decoded model): no permutation gathers
                 identity=bool(perm.shape[0] == n and n > 0 and bool((perm == torch.arange(n, device=dev)).all())))
    try:
        pc._level_cache = cache
    except Exception:
        pass
    return cache


def _plan_key(pc, anchor, mask_anchor_bool):
    return (float(pc.voxel_size), tuple(float(v) for v in pc.level_scale), int(pc.level_num),
            tuple(anchor.shape), mask_anchor_bool is None)


class _LazyChooseMask:
    """The rate subset as the kernels produced it (a list of chosen anchor indices); the bool [N] mask of :1659 is only
    materialised if a level falls off the fused path and asks for it."""

    def __init__(self, n, rows, device):
        self.n, self.rows, self.device, self._t = n, rows, device, None

    def tensor(self):
        if self._t is None:
            self._t = torch.zeros(self.n, dtype=torch.bool, device=self.device)
            self._t[self.rows] = True
        return self._t

    def __getitem__(self, idx):
        return self.tensor()[idx]


def _as_mask(cm):
    return cm.tensor() if isinstance(cm, _LazyChooseMask) else cm


choose_mask_provider = None      # test hook: callable(anchor, mask_anchor_bool) -> bool [N], replaces the random draw of the rate subset


def _choose_args(cache, anchor, mask_anchor_bool, check):
    sizes = cache["sizes"]
    bounds = [0]
    for s_ in sizes:
        bounds.append(bounds[-1] + s_)
    return dict(perm=None if cache.get("identity") else cache["perm"], n=int(anchor.shape[0]), mask=mask_anchor_bool,
                anchor=anchor if check else None, anchor_ref=cache["anchor"] if check else None,
                mask_ref=cache["mask"] if (check and mask_anchor_bool is not None) else None, bounds=bounds)


def begin_step(pc, anchor, mask_anchor_bool, predict_bpp):
    """Enqueue the step's bookkeeping kernel (rate subset + plan check + counts) on the cached plan WITHOUT reading
    its result: the renderer calls this before the read-back of the visible-anchor count it needs anyway, so the
    context model adds no synchronisation point of its own and the GPU does not idle behind it.  The handle is picked
    up by _plan_and_chosen of the same step; anything unexpected (no plan yet, another shape) just returns None."""
    if not (predict_bpp and anchor.is_cuda):
        return None
    cache = getattr(pc, "_level_cache", None)
    if cache is None or pc.level_scale is None or cache["key"] != _plan_key(pc, anchor, mask_anchor_bool) or not cache["covers_all"]:
        return None
    given = choose_mask_provider(anchor, mask_anchor_bool) if choose_mask_provider is not None else None
    seed = _ctx.next_seed() if given is None else 0
    a = _choose_args(cache, anchor, mask_anchor_bool, True)
    h = _ctx.choose_rows_begin(a["perm"], a["n"], a["mask"], given, seed, 0.15, a["anchor"], a["anchor_ref"], a["mask_ref"],
                               a["bounds"])
    packed = None
    if isinstance(pc.latent_codec, _EntropyBottleneck) and pc.latent_codec.filters == (3, 3, 3, 3) and not pc.disable_hyper:
        packed = pc.latent_codec._packed_params()      # a cat launch the step needs anyway: queued before the read-back
    hyper_pre = None
    hl = getattr(pc, "_hyper_latent", None)
    if (packed is not None and FUSED_TRAINING and hl is not None and hl.is_cuda and hl.dtype == torch.float32 and hl.dim() == 2
            and hl.is_contiguous() and hl.shape[0] == a["n"]):
        # the noisy hyper latents in coding order depend on the plan only, not on the rate subset: launched here, in front
        # of the read-back, instead of behind it where the GPU would wait for the host to issue them (a 30 us gap)
        hyper_pre = pc.latent_codec.noisy_latents_launch(hl, None if cache.get("identity") else cache["perm"], _ctx.next_seed())
    return dict(handle=h, cache=cache, seed=seed, given=given, anchor=anchor, mask=mask_anchor_bool, packed=packed,
                hyper_pre=hyper_pre)


def _plan_and_chosen(pc, anchor, mask_anchor_bool, choose_mask, draw=False, begun=None):
    """(_cached_plan(...), per-level row lists of the rate subset, all chosen rows) with ONE host read.

    The cache check ("are anchor and mask what the plan was built from?"), the draw of the subset (when `draw`: the
    15 % of :1658-1659 from the counter-based generator, keyed by the anchor index) and the sizes of the per-level
    subsets are one launch + one read-back; a second launch compacts the chosen rows in coding order
    (ctx_ops.choose_rows).  Per level this equals nonzero(choose_mask[orig]) (:1658-1669 restricted to the level).
    Also leaves the number of live anchors in cache['live_count'] (the mask_anchor_rate of :1661 without a reduction).
    begun: the handle of begin_step() of this step (the first launch is already in flight)."""
    cache = getattr(pc, "_level_cache", None)
    if cache is not None:
        cache.pop("live_count", None)
        cache.pop("_choose_mask", None)
        cache.pop("_nz", None)
        cache.pop("_sub_map", None)
    if (choose_mask is None and not draw) or not anchor.is_cuda:
        return _cached_plan(pc, anchor, mask_anchor_bool), None, None
    fresh = cache is None or cache["key"] != _plan_key(pc, anchor, mask_anchor_bool)
    if fresh:
        cache = _cached_plan(pc, anchor, mask_anchor_bool)               # builds (device sorts + host reads)
    if not cache["covers_all"]:
        return cache, None, None
    use_begun = (begun is not None and not begun.get("used") and not fresh and begun["cache"] is cache and begun["anchor"] is anchor
                 and begun["mask"] is mask_anchor_bool)
    if use_begun:
        begun["used"] = True             # (its handle is read once)
    seed = begun["seed"] if use_begun else (_ctx.next_seed() if draw else 0)
    for attempt in range(2):
        if use_begun and attempt == 0:
            stale, live, per_level, nz, rows, loc, sub_map = _ctx.choose_rows_end(begun["handle"])
        else:
            a = _choose_args(cache, anchor, mask_anchor_bool, not fresh and attempt == 0)
            stale, live, per_level, nz, rows, loc, sub_map = _ctx.choose_rows(
                a["perm"], a["n"], a["mask"], choose_mask, seed, 0.15, a["anchor"], a["anchor_ref"], a["mask_ref"], a["bounds"])
        if not stale:
            break
        cache = _cached_plan(pc, anchor, mask_anchor_bool)               # rebuilds (rare: anchors / anchor mask changed)
        fresh = True
    n = int(anchor.shape[0])
    sizes = cache["sizes"]
    cache["live_count"] = live if mask_anchor_bool is not None else n
    cache["_nz"] = nz                       # coding-order positions of the chosen rows (this step)
    cache["_sub_map"] = sub_map             # per level: row -> index in the level's chosen list, -1 if not chosen
    cum = [0]
    for v in per_level:
        cum.append(cum[-1] + v)
    locs = [loc[cum[j]:cum[j + 1]] for j in range(len(sizes))]
    return cache, locs, rows


def _level_plan_uncached(pc, anchor, mask_anchor_bool):
    _hl, inverse_indices_list, mapping_list, _ = divide_levels(pc, anchor, mask_anchor_bool)
    n = anchor.shape[0]
    dev = anchor.device
    plan = []
    for i in reversed(range(pc.level_num)):
        n_level = n if i == 0 else mapping_list[i - 1].shape[0]
        to_code = torch.ones(n_level, dtype=torch.bool, device=dev)
        if i != pc.level_num - 1:
            to_code[mapping_list[i]] = False
        if i != 0:
            orig = mapping_to_orign(mapping_list, i, to_code)
        else:
            orig = torch.arange(n, device=dev)[to_code]
        plan.append((i, to_code, orig, None))     # level anchors are re-gathered from the live tensor: level_anchors()
    return plan, inverse_indices_list, mapping_list


def level_anchors(anchor, mask_anchor_bool, level, orig):
    """hybrid_anchor_list[level][to_code] of the reference (:1594): the rows `orig` of the anchors the level
    was built from (masked anchors are zeroed from level 1 up, :1758-1759)."""
    if level >= 1 and mask_anchor_bool is not None:
        anchor = anchor * mask_anchor_bool.unsqueeze(1)
    return gather_unique(anchor, orig)


def grid_mlp(pc, level, feat_in):
    """mlp_grid[level](feat_in) (:1600) on the fused fp32-MFMA kernels when the shape has an instance."""
    seq = pc.get_grid_mlp[level]
    return _mlp.mlp2(feat_in, seq) if _mlp.supported(seq) else seq(feat_in)


def split_prediction(pc, predicted):                                    # :1603-1608
    D, K = pc.feat_dim, pc.n_offsets
    (mean_feat, scale_feat, mean_scaling, scale_scaling, mean_offsets, scale_offsets, qf, qs, qo) = torch.split(
        predicted, [D, D, 6, 6, 3 * K, 3 * K, 1, 1, 1], dim=-1)
    Q_feat = (Q_FEAT0 * (1 + torch.tanh(qf))).clamp(1e-9)
    Q_scaling = (Q_SCALING0 * (1 + torch.tanh(qs))).clamp(1e-9)
    Q_offsets = (Q_OFFSETS0 * (1 + torch.tanh(qo))).clamp(1e-9)
    return mean_feat, scale_feat, mean_scaling, scale_scaling, mean_offsets, scale_offsets, Q_feat, Q_scaling, Q_offsets


def _index_rows(x, idx):
    """x.index_select(0, idx), through cgs_gather_rows for narrow fp32 rows on the device (one lane, or a float2 / float4 lane,
    per row piece instead of torch's generic gather kernel: 29 -> ~8 us at 1 M rows of 3 floats; csrc/ctx.hip)."""
    n = int(idx.shape[0])
    w = int(x[0].numel()) if x.shape[0] > 0 else 0
    if (x.is_cuda and x.dtype == torch.float32 and idx.dtype == torch.int64 and x.is_contiguous() and idx.is_contiguous()
            and 1 <= w <= 64 and n > 0):
        out = torch.empty((n,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().cgs_gather_rows(_lib.ptr(x), _lib.ptr(idx), n, w, _lib.ptr(out), _lib.current_stream()),
                   "cgs_gather_rows")
        return out
    if (x.is_cuda and x.dtype == torch.int64 and x.dim() == 1 and idx.dtype == torch.int64 and x.is_contiguous() and idx.is_contiguous()
            and n > 0 and x.shape[0] > 0):
        # an index vector through an index vector (inv_perm[vis_idx]): the same kernel on the rows' bits, two float lanes per entry
        # (plain moves: every bit pattern survives) instead of torch's generic index kernel (10 -> 6 us at 1 M entries)
        out = torch.empty(n, dtype=torch.int64, device=x.device)
        _lib.check(_lib.lib().cgs_gather_rows(_lib.ptr(x), _lib.ptr(idx), n, 2, _lib.ptr(out), _lib.current_stream()), "cgs_gather_rows")
        return out
    return x.index_select(0, idx)


class _GatherUnique(torch.autograd.Function):
    """x[idx] for UNIQUE row indices: the backward is a plain row scatter into zeros (index_copy_) instead of
    torch's index_put_(accumulate=True), which sorts the indices first (rocprof: ~44 ms of merge-sort +
    indexing_backward kernels per 14 steps at 1 M anchors)."""

    @staticmethod
    def forward(ctx, x, idx, complete=False, values=None):
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        ctx.complete = complete and idx.shape[0] == x.shape[0]
        ctx.ascending = bool(getattr(idx, "_cgs_ascending", False))
        if values is not None:          # the rows were gathered earlier in the forward (gather_unique_attach)
            return values.view_as(values)
        return _index_rows(x.detach(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        n, N = int(idx.shape[0]), int(ctx.shape[0])
        if (ctx.ascending and not ctx.complete and g.is_cuda and g.dtype == torch.float32 and N > 0 and 8 * n >= N
                and idx.dtype == torch.int64 and 1 <= int(g[0].numel()) <= 256):
            # ascending rows (the visible-anchor list): zero fill and scatter in ONE pass (cgs_scatter_rows_sorted)
            g = g.contiguous()
            out = torch.empty(ctx.shape, dtype=g.dtype, device=g.device)
            _lib.check(_lib.lib().cgs_scatter_rows_sorted(_lib.ptr(g), _lib.ptr(idx), n, N, int(out[0].numel()), _lib.ptr(out),
                                                         _lib.current_stream()), "cgs_scatter_rows_sorted")
            return out, None, None, None
        # `complete`: idx is a permutation of all rows, every row is written, no zero fill needed
        out = (torch.empty if ctx.complete else torch.zeros)(ctx.shape, dtype=g.dtype, device=g.device)
        out.index_copy_(0, idx, g.contiguous())
        return out, None, None, None


class _GatherUniquePair(torch.autograd.Function):
    """(x[idx_a], x.reshape(N, -1)[idx_b]) — two lists of distinct rows of the SAME tensor, possibly overlapping — as ONE node:
    the backward writes one gradient buffer (zero fill + scatter of the first list in one pass when it is ascending, the second
    list's rows added on top) instead of two full-size buffers and the engine's add of the two.  The mask weights' two readers
    (the visible anchors' rows for the expansion, gaussian_renderer/__init__.py:80, and the rate subset's rows,
    scene/gaussian_model.py:1664-1669): fill [N,K] + index_copy_ + add [N,K,1] = 38 us -> one row-add of 15 % of the rows.
    vals_a: the first list's rows gathered earlier in the forward (see gather_unique_attach), or None."""

    @staticmethod
    def forward(ctx, x, idx_a, vals_a, idx_b):
        ctx.save_for_backward(idx_a, idx_b)
        ctx.shape = x.shape
        ctx.ascending = bool(getattr(idx_a, "_cgs_ascending", False))
        ctx.set_materialize_grads(False)
        xd = x.detach()
        xa = vals_a.view_as(vals_a) if vals_a is not None else _index_rows(xd, idx_a)
        return xa, _index_rows(xd.reshape(xd.shape[0], -1), idx_b)

    @staticmethod
    def backward(ctx, ga, gb):
        idx_a, idx_b = ctx.saved_tensors
        if ga is None and gb is None:
            return None, None, None, None
        like = ga if ga is not None else gb
        N = int(ctx.shape[0])
        w = 1
        for d in ctx.shape[1:]:
            w *= int(d)
        n = int(idx_a.shape[0])
        if (ga is not None and ctx.ascending and ga.is_cuda and ga.dtype == torch.float32 and N > 0 and 8 * n >= N
                and idx_a.dtype == torch.int64 and 1 <= w <= 256):
            ga = ga.contiguous()
            out = torch.empty(ctx.shape, dtype=ga.dtype, device=ga.device)
            _lib.check(_lib.lib().cgs_scatter_rows_sorted(_lib.ptr(ga), _lib.ptr(idx_a), n, N, w, _lib.ptr(out),
                                                         _lib.current_stream()), "cgs_scatter_rows_sorted")
        else:
            out = torch.zeros(ctx.shape, dtype=like.dtype, device=like.device)
            if ga is not None:
                out.index_copy_(0, idx_a, ga.contiguous())
        if gb is not None:
            _ctx.add_rows_(out, idx_b, gb)           # distinct rows: one add per element, no order to depend on
        return out, None, None, None


def gather_unique_pair_attach(x, idx_a, vals_a, idx_b):
    """See _GatherUniquePair."""
    return _GatherUniquePair.apply(x, idx_a, vals_a, idx_b)


class _JoinRows(torch.autograd.Function):
    """The row slices `parts` (consecutive, covering `whole`) were written in place by the fused kernels:
    return `whole` as their concatenation without copying; the backward hands each part a view of the gradient."""

    @staticmethod
    def forward(ctx, whole, *parts):
        ctx.sizes = [int(p.shape[0]) for p in parts]
        assert sum(ctx.sizes) == whole.shape[0]
        ctx.set_materialize_grads(False)     # (a context level that summed its prefix gradient in place returns none: no zero tensor for it)
        return whole.view_as(whole)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * (1 + len(ctx.sizes))
        g = g.contiguous()
        _ctx.note_joined_grad(g)        # (a context level may add its prefix gradient into the rows in front of its own: ctx_ops._grad_prefix)
        return (None, *torch.split(g, ctx.sizes))


def _rowcat_ok(x):
    # torch's index_select is the faster row gather (tools/gather_micro.py) EXCEPT for rows that are a multiple of
    # 16 bytes, where it switches to a vectorized_gather kernel that runs ~6x slower on gfx950 (214 vs 33 us for
    # 1 M rows of 12 floats): those go through the one-source rowcat kernel.
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2 and x.shape[0] > 0 and x[0].numel() > 0
            and (x[0].numel() * 4) % 16 == 0)


def gather_unique(x, idx, complete=False):
    """x[idx] for distinct rows idx; complete = idx is a permutation of ALL rows of x."""
    if _rowcat_ok(x):                # one-source rowcat: row gather forward, plain row scatter backward (HIP)
        return _ctx.rowcat([(x.reshape(x.shape[0], -1), idx, True)]).view((idx.shape[0],) + tuple(x.shape[1:]))
    return _GatherUnique.apply(x, idx, complete) if x.requires_grad else _index_rows(x, idx)


def gather_unique_attach(x, idx, values):
    """The autograd node of gather_unique(x, idx) around rows that were gathered EARLIER (values = _index_rows(x.detach(), idx)):
    the launch stays where it hides a host wait, the node is created where the backward should run it (see
    ctx_ops._MaskSTEAttach)."""
    return _GatherUnique.apply(x, idx, False, values) if x.requires_grad else values


class _GatherRows(torch.autograd.Function):
    """x[idx] for arbitrary (repeating) row indices with an atomic scatter-add backward (index_add_) instead of
    torch's sort-based index_put_(accumulate=True)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        out = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        out.index_add_(0, idx, g.contiguous())
        return out, None


def gather_rows(x, idx):
    if _rowcat_ok(x):
        return _ctx.rowcat([(x.reshape(x.shape[0], -1), idx, False)]).view((idx.shape[0],) + tuple(x.shape[1:]))
    return _GatherRows.apply(x, idx) if x.requires_grad else x.index_select(0, idx)


class _EarlyMismatch(Exception):
    """The level kernels were enqueued ahead of the rate subset's read-back and what came back does not fit them (the plan was
    rebuilt, or a level left the fused path): the caller runs the step's context model again the plain way."""


def _early_levels(pc, anchor, hyper, feat, grid_offsets, grid_scaling, mask_anchor_bool, begun):
    """Enqueue the step's level kernels (cgs_ctx_level_fwd, one per level) NOW, in front of the read-back of the rate subset's
    counts: they read the anchors, the parameters, the noisy hyper latents begin_step() launched and the plan — nothing the host
    has to wait for — and take ~0.4 ms of the device, which used to sit idle for ~130 us in three gaps while the host, just
    back from the read-back, prepared them one after the other (profiles/r06_gpu_idle_gaps.txt).  The autograd nodes are created
    later, by the level loop, around these results (ctx_ops.level_fused(pre=)).  None when the step is not the all-fused one."""
    c = begun["cache"]
    K, D = pc.n_offsets, pc.feat_dim
    if (begun.get("used") or c is not getattr(pc, "_level_cache", None) or pc.level_scale is None
            or c["key"] != _plan_key(pc, anchor, mask_anchor_bool) or not c["covers_all"] or c.get("identity")
            or begun["anchor"] is not anchor or begun["mask"] is not mask_anchor_bool):
        return None
    if not (FUSED_TRAINING and ROW_SOURCE and LEVEL_FUSED and RATE_FUSED and RATE_SIDE and not pc.adaptQ_per_channel):
        return None
    pre = begun.get("hyper_pre")
    if (pre is None or pre[0] is not hyper or pc.disable_hyper or hyper.dtype != torch.float32
            or not (isinstance(pc.latent_codec, _EntropyBottleneck) and pc.latent_codec.filters == (3, 3, 3, 3))):
        return None
    if not (grid_offsets.dim() == 3 and pc._anchor_feat.is_cuda
            and (feat.requires_grad or grid_scaling.requires_grad or grid_offsets.requires_grad)):
        return None
    sizes, perm, plan = c["sizes"], c["perm"], c["plan"]
    if any(n_ <= 0 for n_ in sizes) or not all(_mlp.supported(pc.get_grid_mlp[i_]) for (i_, _t, _o, _a) in plan):
        return None
    v_blocks = torch.split(pre[1], sizes)
    row_src = _ctx.RowSource(feat, grid_scaling, grid_offsets, True)
    if not all(_ctx.level_fused_supported(pc.get_grid_mlp[i_], anchor, v_blocks[j_], row_src) for j_, (i_, _t, _o, _a) in enumerate(plan)):
        return None
    row_src.sums_buffer()
    n_tot = int(perm.shape[0])
    dev = anchor.device
    big_f = torch.empty(n_tot, feat.shape[1], dtype=torch.float32, device=dev)
    big_s = torch.empty(n_tot, grid_scaling.shape[1], dtype=torch.float32, device=dev)
    big_o = torch.empty(n_tot, 3 * K, dtype=torch.float32, device=dev)
    out, row_off, src_vals = [], 0, None
    for j, (i, _tc, orig, _a) in enumerate(plan):
        n_l = sizes[j]
        sl = slice(row_off, row_off + n_l)
        if src_vals is None:
            a_rows, pos_, bf, bs = orig, None, None, None
            a_mask = mask_anchor_bool if (i >= 1 and mask_anchor_bool is not None) else None
        else:
            (a_rows, pos_, bf, bs), a_mask = src_vals, None
        seed = _ctx.next_seed()
        res = _ctx.level_fused_launch(anchor, bf, bs, v_blocks[j], pc.get_grid_mlp[i], 2 * (D + 6 + 3 * K), a_rows, a_mask, pos_,
                                      row_src, perm[row_off:row_off + n_l], (big_f[sl], big_s[sl], big_o[sl]),
                                      (Q_FEAT0, Q_SCALING0, Q_OFFSETS0), seed)
        out.append((res, seed))
        row_off += n_l
        if i != 0:          # (what _next_context hands the next level, as values: the coded prefix lies at the front of the buffers)
            src_vals = (c["ctx_idx"][i], c["ctx_pos"][i], big_f[:row_off], big_s[:row_off])
    return dict(cache=c, row_src=row_src, big=(big_f, big_s, big_o), levels=out, inputs=(hyper, feat, grid_offsets, grid_scaling))


def early_levels_begin(pc, anchor, hyper, feat, grid_offsets, grid_scaling, mask_anchor_bool, begun):
    """_early_levels for a caller that knows it will run the training context model on exactly these tensors: the renderer calls it
    right behind begin_step(), i.e. even before it reads the visible-anchor count — the level kernels do not depend on the view.
    The result waits in begun["early"] for context_model_coding_order."""
    if EARLY_LEVELS and begun is not None and anchor.is_cuda and torch.is_grad_enabled():
        begun["early"] = _early_levels(pc, anchor, hyper, feat, grid_offsets, grid_scaling, mask_anchor_bool, begun)


def context_model_coding_order(pc, anchor, hyper, feat, grid_offsets, grid_scaling, mask_anchor_bool, training,
                               keep_stats, choose_mask=None, draw_choose=False, begun=None, allow_rate_lazy=False):
    """_coding_order_impl with the level kernels enqueued ahead of the rate subset's read-back when the step allows it
    (_early_levels); if what comes back does not fit them, once more the plain way."""
    args = (pc, anchor, hyper, feat, grid_offsets, grid_scaling, mask_anchor_bool, training, keep_stats)
    try:
        return _coding_order_impl(*args, choose_mask=choose_mask, draw_choose=draw_choose, begun=begun,
                                  allow_rate_lazy=allow_rate_lazy, early_ok=EARLY_LEVELS)
    except _EarlyMismatch:
        if begun is not None:
            begun["used"] = True
        return _coding_order_impl(*args, choose_mask=choose_mask, draw_choose=draw_choose, begun=begun,
                                  allow_rate_lazy=allow_rate_lazy, early_ok=False)


def _coding_order_impl(pc, anchor, hyper, feat, grid_offsets, grid_scaling, mask_anchor_bool, training,
                       keep_stats, choose_mask=None, draw_choose=False, begun=None, allow_rate_lazy=False, early_ok=False):
    """The level loop of multi_scale_generating (:1556-1652) in coding order.

    Returns (cache, feat_Q, scaling_Q, offsets_Q [rows in coding order: row r is anchor cache['perm'][r]],
    likelihood_hyper [N-ordered], levels) where `levels` holds, per coded level, the slices the rate model
    needs (only when keep_stats)."""
    K = pc.n_offsets
    _ctx._JOINED_GRADS.clear()
    if pc.level_scale is None:                                                         # :1559
        sel = anchor[mask_anchor_bool] if mask_anchor_bool is not None else anchor
        pc.level_scale = find_divide_scale(pc, sel, pc.target_ratio, pc.level_num)
    early = None
    if (early_ok and begun is not None and training and keep_stats and allow_rate_lazy and anchor.is_cuda
            and (draw_choose or choose_mask is not None)):
        early = begun.pop("early", None)
        if early is not None and not all(a_ is b_ for a_, b_ in zip(early["inputs"], (hyper, feat, grid_offsets, grid_scaling))):
            early = None                    # (launched for other tensors: its buffers are simply dropped)
        if early is None:
            early = _early_levels(pc, anchor, hyper, feat, grid_offsets, grid_scaling, mask_anchor_bool, begun)
    # level plan (cached) + the rate subset's rows per level, with one host read for both
    c, locs, chosen_rows = _plan_and_chosen(pc, anchor, mask_anchor_bool,
                                            choose_mask if (keep_stats and anchor.is_cuda) else None,
                                            draw=draw_choose and keep_stats and choose_mask is None, begun=begun)
    if early is not None and (c is not early["cache"] or locs is None or chosen_rows is None):
        raise _EarlyMismatch()
    if draw_choose and keep_stats and choose_mask is None:
        # the kernels drew the subset: keep a bool mask around for the (rare) level that does not take the fused path
        if chosen_rows is None:            # anchors the plan does not cover: the torch draw
            choose_mask = draw_choose_mask(anchor, mask_anchor_bool, False)
        else:
            choose_mask = _LazyChooseMask(int(anchor.shape[0]), chosen_rows, anchor.device)
        c["_choose_mask"] = choose_mask
    perm, sizes = c["perm"], c["sizes"]
    # (host work that launches nothing — the row source, the level outputs' buffers, the path predicates — comes BEFORE the hyper
    #  step's launches: the device is idle behind the rate subset's read-back at this point and whatever the host still has to do
    #  between the hyper step's ~95 us kernel and the first level kernel shows as a gap; profiles/r06_gpu_idle_gaps.txt)
    # one gather per tensor into coding order, then contiguous per-level slices (split backward = one cat)
    full = c["covers_all"]
    if c.get("identity"):       # parameters are stored in coding order: the level slices are views, nothing moves
        in_order = lambda t: t
    else:
        in_order = lambda t: gather_unique(t, perm, full)
    # training path on the device: the level's element-wise stages run as fused HIP kernels (ctx_ops)
    fused = FUSED_TRAINING and training and anchor.is_cuda and not pc.adaptQ_per_channel
    # When every level takes the fused path, features / scaling / offsets are not gathered at all: the level kernels
    # read their rows of the parameter tensors through perm[lo:hi] and scatter the gradients back (ctx_ops.RowSource)
    row_src = None
    if (fused and ROW_SOURCE and not c.get("identity") and (not keep_stats or choose_mask is not None)
            and all(_mlp.supported(pc.get_grid_mlp[i_]) for (i_, _t, _o, _a) in c["plan"])
            and grid_offsets.dim() == 3 and (feat.requires_grad or grid_scaling.requires_grad or grid_offsets.requires_grad)):
        if early is not None:
            row_src = early["row_src"]
        else:
            row_src = _ctx.RowSource(feat, grid_scaling, grid_offsets, full)
            row_src.sums_buffer()            # (its zero fill is queued here, in front of the hyper step's kernel)
        feat_l = scal_l = off_l = None
    else:
        if early is not None:
            raise _EarlyMismatch()
        feat_l = torch.split(in_order(feat), sizes)
        scal_l = torch.split(in_order(grid_scaling), sizes)
        off_l = torch.split(in_order(grid_offsets), sizes)
    n_tot = int(perm.shape[0])
    if early is not None:
        big_f, big_s, big_o = early["big"]
    elif fused:
        big_f = torch.empty(n_tot, feat.shape[1], dtype=torch.float32, device=anchor.device)
        big_s = torch.empty(n_tot, grid_scaling.shape[1], dtype=torch.float32, device=anchor.device)
        big_o = torch.empty(n_tot, 3 * K, dtype=torch.float32, device=anchor.device)
    # :1556.  Only the rate subset's hyper likelihood is ever read (:1662), and only as a sum of bits: on the fused
    # training path the bottleneck returns the noisy latents already in coding order plus that sum (two launches);
    # otherwise the likelihood of the chosen rows / of all rows as a tensor
    hyp_p = None
    eb_mine = isinstance(pc.latent_codec, _EntropyBottleneck) and pc.latent_codec.filters == (3, 3, 3, 3)
    if (FUSED_TRAINING and training and keep_stats and choose_mask is not None and hyper.is_cuda and eb_mine
            and not pc.disable_hyper and c.get("_nz") is not None and c["covers_all"] and hyper.dtype == torch.float32):
        pre = begun.get("hyper_pre") if (begun is not None and begun.get("cache") is c) else None
        if pre is not None and pre[0] is not hyper:
            pre = None
        hyp_p, likelihood_hyper = pc.latent_codec.training_step_forms(
            hyper, None if c.get("identity") else perm, None if c.get("identity") else c["inv_perm"], c["_nz"],
            pre[2] if pre is not None else _ctx.next_seed(), packed=begun.get("packed") if begun is not None else None,
            noisy=pre[1] if pre is not None else None, sizes=sizes if (HYPER_BLOCKS and len(sizes) <= 4) else None,
            rows_orig=chosen_rows)
        hyper_feat = None
    elif FUSED_TRAINING and training and keep_stats and choose_mask is not None and hyper.is_cuda and eb_mine:
        rows_h = chosen_rows if chosen_rows is not None else torch.nonzero(_as_mask(choose_mask))[:, 0]
        hyper_feat, likelihood_hyper = pc.latent_codec(hyper, training=training, rows=rows_h)
    else:
        hyper_feat, likelihood_hyper = pc.latent_codec(hyper, training=training)
    if pc.disable_hyper and hyper_feat is not None:
        hyper_feat = hyper_feat * 0

    hyp_l = hyp_p if isinstance(hyp_p, tuple) else torch.split(hyp_p if hyp_p is not None else in_order(hyper_feat), sizes)

    feat_q, scal_q, off_q, levels = [], [], [], []
    ctx_src = None                      # (idx, pos, base_f, base_s): the coded context of the next level
    # the fused levels write their outputs side by side into these (coding order), so no cat is needed at the end
    row_off, joined = 0, fused
    # The mean / scale outputs of the rate subset can stay INSIDE the rate kernels (cgs_rate_sub_*, round 6) when the rate model
    # will take its one-node path (rate_model: every level fused, the subset listed level by level, the live fraction a host
    # number) — decided here, before the first level runs, from the same predicates the loop and rate_model use
    rate_lazy = bool(
        allow_rate_lazy and RATE_FUSED and fused and LEVEL_FUSED and RATE_SIDE and keep_stats and choose_mask is not None
        and chosen_rows is not None and locs is not None and row_src is not None
        and (mask_anchor_bool is None or c.get("live_count") is not None) and pc._anchor_feat.is_cuda
        and all(sizes[j_] == 0 or (_mlp.supported(pc.get_grid_mlp[i_])
                                   and _ctx.level_fused_supported(pc.get_grid_mlp[i_], anchor, hyp_l[j_], row_src))
                for j_, (i_, _t, _o, _a2) in enumerate(c["plan"])))
    for j, (i, _tc, orig, _a) in enumerate(c["plan"]):
        n_l = sizes[j]
        if n_l > 0:
            loc = None
            if keep_stats and choose_mask is not None:
                loc = locs[j] if locs is not None else torch.nonzero(choose_mask[orig])[:, 0]
            subset_mode = _mlp.supported(pc.get_grid_mlp[i]) and (not keep_stats or loc is not None)
            use_fused = fused and subset_mode
            if (use_fused and LEVEL_FUSED and row_src is not None and joined
                    and _ctx.level_fused_supported(pc.get_grid_mlp[i], anchor, hyp_l[j], row_src)):
                # input-row assembly + hidden layer + step sizes + noisy values of every row: one launch each way
                # (csrc/ctx_level.hip); the mean / scale outputs only on the rate subset's rows
                sl = slice(row_off, row_off + n_l)
                side = _ctx.RateSide() if (keep_stats and RATE_SIDE) else None
                if ctx_src is None:
                    # (masked anchors sit at the origin from level 1 up, :1758-1759: the product is folded into the gather)
                    a_rows, pos_, base_f, base_s, csr = orig, None, None, None, None
                    a_mask = mask_anchor_bool if (i >= 1 and mask_anchor_bool is not None) else None
                else:
                    a_rows, pos_, base_f, base_s, csr = ctx_src
                    a_mask = None
                pre_j, seed_j = early["levels"][j] if early is not None else (None, None)
                hf, hs, ho, Q_all, pred_sub = _ctx.level_fused(
                    anchor, base_f, base_s, hyp_l[j], pc.get_grid_mlp[i], 2 * (pc.feat_dim + 6 + 3 * K), loc if keep_stats else None,
                    a_rows, a_mask, pos_, csr, row_src, perm[row_off:row_off + n_l], (big_f[sl], big_s[sl], big_o[sl]), side,
                    (Q_FEAT0, Q_SCALING0, Q_OFFSETS0), seed=seed_j, rate_lazy=rate_lazy and keep_stats, pre=pre_j)
                row_off += n_l
                if keep_stats:
                    span = None
                    if chosen_rows is not None:
                        lo_ = sum(int(l_.shape[0]) for l_ in locs[:j])
                        span = (chosen_rows, lo_, lo_ + int(loc.shape[0]))
                    sm = c.get("_sub_map")
                    lvl0 = sum(sizes[:j])
                    levels.append(dict(level=i, orig=orig, rows=orig[loc] if span is None else None, loc=loc, n_level=n_l,
                                       fused=True, yf=hf, ys=hs, yo=ho, Q=Q_all, chosen=span, side=side, side_src=row_src,
                                       sub_map=sm[lvl0:lvl0 + n_l] if (sm is not None and span is not None) else None,
                                       pred=pred_sub, lazy=rate_lazy))
                feat_q.append(hf)
                scal_q.append(hs)
                off_q.append(ho)
                if i != 0:
                    ctx_src = _next_context(c, i, feat_q, scal_q, (big_f, big_s, row_off))
                continue
            if early is not None:
                raise _EarlyMismatch()
            if rate_lazy:
                raise RuntimeError("context model: a level left the fused path after the rate subset was made lazy")
            if ctx_src is None:                                                        # :1596-1600
                if use_fused:
                    # (masked anchors sit at the origin from level 1 up, :1758-1759: the product is folded into the gather)
                    a_mask = mask_anchor_bool if (i >= 1 and mask_anchor_bool is not None) else None
                    feat_in = _ctx.rowcat([(anchor, orig, True, a_mask), (hyp_l[j], None, True)])
                else:
                    feat_in = torch.cat([level_anchors(anchor, mask_anchor_bool, i, orig), hyp_l[j].float()], dim=1)
            else:
                idx, pos, base_f, base_s, csr = ctx_src
                if use_fused:
                    feat_in = _ctx.ctx_assemble(anchor, base_f, base_s, hyp_l[j], idx, pos, csr)
                else:
                    feat_in = torch.cat([gather_rows(anchor, idx), gather_rows(base_f, pos), gather_rows(base_s, pos),
                                         hyp_l[j]], dim=1)
            # Only the three step-size outputs of mlp_grid are needed for EVERY row (they scale the noise /
            # the rounding); the 172 mean/scale outputs are consumed by the rate model alone, i.e. by the
            # `choose_mask` rows (15 % in training, :1658-1669; none at all when predict_bpp is off).  The
            # reference evaluates all 175 outputs for all rows and throws 85-100 % of them away; here the
            # second layer runs with its 3 step-size rows on every anchor and with all rows on the chosen
            # anchors only (identical values: the same fp32 fma chains).
            feat_sub = pred_sub = None
            if subset_mode:
                seq = pc.get_grid_mlp[i]
                D_ = pc.feat_dim
                n_stat = 2 * (D_ + 6 + 3 * K)
                if use_fused and keep_stats:
                    # mlp_grid has two consumers, the step sizes on every row and all outputs on the chosen rows: ONE
                    # autograd node serves both (weight gradients of the module filled once, the subset's input
                    # gradient merged row-wise into the full one)
                    qadj, pred_sub = _mlp.level_mlp(feat_in, loc, seq, n_stat)
                else:
                    qadj = _mlp.mlp2_weights(feat_in, seq[0].weight, seq[0].bias, seq[2].weight[n_stat:], seq[2].bias[n_stat:])
            if use_fused:
                # step sizes + noise (:1603-1616) in one launch; the rate of the chosen rows is one more (rate_model)
                sl = slice(row_off, row_off + n_l)
                outs = (big_f[sl], big_s[sl], big_o[sl]) if joined else None
                side = None
                if row_src is not None:
                    side = _ctx.RateSide() if (keep_stats and RATE_SIDE) else None
                    hf, hs, ho, Q_all = _ctx.noise_quant(None, None, None, qadj, (Q_FEAT0, Q_SCALING0, Q_OFFSETS0),
                                                         outs=outs, src=row_src, rows=perm[row_off:row_off + n_l], side=side)
                else:
                    hf, hs, ho, Q_all = _ctx.noise_quant(feat_l[j], scal_l[j], off_l[j].reshape(n_l, 3 * K), qadj,
                                                         (Q_FEAT0, Q_SCALING0, Q_OFFSETS0), outs=outs)
                row_off += n_l
                if keep_stats:
                    # chosen_rows lists the chosen anchors level by level: this level's are [lo, lo + len(loc))
                    span = None
                    if chosen_rows is not None:
                        lo_ = sum(int(l_.shape[0]) for l_ in locs[:j])
                        span = (chosen_rows, lo_, lo_ + int(loc.shape[0]))
                    sm = c.get("_sub_map")
                    lvl0 = sum(sizes[:j])
                    # `rows` (original indices of the chosen anchors) is only read when the subset is not listed level by
                    # level (span None): otherwise the mask rows come from ONE gather over all levels' chosen anchors
                    levels.append(dict(level=i, orig=orig, rows=orig[loc] if span is None else None, loc=loc, n_level=n_l,
                                       fused=True, yf=hf, ys=hs,
                                       yo=ho, Q=Q_all, chosen=span, side=side, side_src=row_src,
                                       sub_map=sm[lvl0:lvl0 + n_l] if (sm is not None and span is not None) else None,
                                       pred=pred_sub))
                feat_q.append(hf)
                scal_q.append(hs)
                off_q.append(ho)
                if i != 0:
                    ctx_src = _next_context(c, i, feat_q, scal_q, (big_f, big_s, row_off) if joined else None)
                continue
            joined = False
            if subset_mode:
                Q_feat = (Q_FEAT0 * (1 + torch.tanh(qadj[:, 0:1]))).clamp(1e-9)
                Q_scaling = (Q_SCALING0 * (1 + torch.tanh(qadj[:, 1:2]))).clamp(1e-9)
                Q_offsets = (Q_OFFSETS0 * (1 + torch.tanh(qadj[:, 2:3]))).clamp(1e-9)
            else:
                (mean_feat, scale_feat, mean_scaling, scale_scaling, mean_offsets, scale_offsets, Q_feat, Q_scaling,
                 Q_offsets) = split_prediction(pc, grid_mlp(pc, i, feat_in))
            hf, hs, ho = feat_l[j], scal_l[j], off_l[j]
            qo = Q_offsets.view(n_l, K, -1) if pc.adaptQ_per_channel else Q_offsets.unsqueeze(1)
            if training:                                                               # :1610-1616
                hf = hf + torch.empty_like(hf).uniform_(-0.5, 0.5) * Q_feat
                hs = hs + torch.empty_like(hs).uniform_(-0.5, 0.5) * Q_scaling
                ho = ho + torch.empty_like(ho).uniform_(-0.5, 0.5) * qo
            else:                                                                      # :1617-1625
                hf = STE_multistep.apply(hf, Q_feat).detach()
                hs = STE_multistep.apply(hs, Q_scaling).detach()
                ho = STE_multistep.apply(ho, qo).detach()
            ho = ho.reshape(-1, 3 * K)
            if keep_stats and subset_mode:
                g = lambda t: gather_unique(t, loc)
                (mean_feat, scale_feat, mean_scaling, scale_scaling, mean_offsets, scale_offsets, _qf, _qs, _qo) = \
                    split_prediction(pc, grid_mlp(pc, i, g(feat_in)))
                levels.append(dict(level=i, orig=orig, rows=orig[loc], n_level=n_l, selected=True, feat=g(hf), scaling=g(hs),
                                   offsets=g(ho), mf=mean_feat, sf=scale_feat, qf=g(Q_feat), ms=mean_scaling,
                                   ss=scale_scaling, qs=g(Q_scaling), mo=mean_offsets, so=scale_offsets, qo=g(Q_offsets)))
            elif keep_stats:
                levels.append(dict(level=i, orig=orig, n_level=n_l, selected=False, feat=hf, scaling=hs, offsets=ho,
                                   mf=mean_feat, sf=scale_feat, qf=Q_feat, ms=mean_scaling, ss=scale_scaling,
                                   qs=Q_scaling, mo=mean_offsets, so=scale_offsets, qo=Q_offsets))
        else:
            if feat_l is not None:
                hf, hs, ho = feat_l[j], scal_l[j], off_l[j].reshape(-1, 3 * K)
            else:       # an EMPTY level on the row-source path (tiny scenes): the per-level slices were never materialised
                hf, hs, ho = (anchor.new_zeros(0, w, dtype=torch.float32) for w in (pc.feat_dim, 6, 3 * K))
            joined = joined and n_l == 0
        feat_q.append(hf)
        scal_q.append(hs)
        off_q.append(ho)
        if i != 0:
            ctx_src = _next_context(c, i, feat_q, scal_q)
    if joined and row_off == n_tot:
        return (c, _JoinRows.apply(big_f, *feat_q), _JoinRows.apply(big_s, *scal_q),
                _JoinRows.apply(big_o, *off_q).view(-1, K, 3), likelihood_hyper, levels)
    cat = lambda parts: parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
    return c, cat(feat_q), cat(scal_q), cat(off_q).view(-1, K, 3), likelihood_hyper, levels


def _next_context(c, i, feat_q, scal_q, joined=None):
    """:1650-1651 / 1711-1724 — what the level coded after level i reads of the already coded anchors: their
    original rows (for the anchor position) and their positions in the coded prefix (<= 20 % of N)."""
    if len(feat_q) == 1:
        base_f, base_s = feat_q[0], scal_q[0]
    elif joined is not None:            # the prefix already lies contiguously in the level output buffers
        big_f, big_s, rows = joined
        base_f, base_s = _JoinRows.apply(big_f[:rows], *feat_q), _JoinRows.apply(big_s[:rows], *scal_q)
    else:
        base_f, base_s = torch.cat(feat_q, dim=0), torch.cat(scal_q, dim=0)
    return c["ctx_idx"][i], c["ctx_pos"][i], base_f, base_s, c["ctx_csr"][i]


def draw_choose_mask(anchor, mask_anchor_bool, return_sum_bits):
    """:1658-1661 — the anchors whose bits enter the rate estimate."""
    thresh = 1 if return_sum_bits else 0.15
    choose_mask = torch.rand_like(anchor[:, 0]) <= thresh
    if mask_anchor_bool is not None:
        choose_mask = choose_mask & mask_anchor_bool
    return choose_mask


def rate_model(pc, anchor, binary_grid_masks, mask_anchor_bool, likelihood_hyper, levels, return_sum_bits,
               choose_mask=None, live_count=None, masks_chosen_given=None):
    """:1657-1707 — bits of a random 15 % subset (all anchors for return_sum_bits), per level on the level's rows.
    masks_chosen_given: binary_grid_masks.reshape(n, K)[levels[0]["chosen"][0]] gathered by the caller (one node with the
    visible rows' gather, _GatherUniquePair), or None."""
    K = pc.n_offsets
    n = anchor.shape[0]
    dev = anchor.device
    if choose_mask is None:
        choose_mask = draw_choose_mask(anchor, mask_anchor_bool, return_sum_bits)
    if mask_anchor_bool is None:
        mask_anchor_rate = 1
    elif live_count is not None:                # counted by the step's bookkeeping kernel: no reduction, a host scalar
        mask_anchor_rate = live_count / mask_anchor_bool.numel()
    else:
        mask_anchor_rate = (mask_anchor_bool.sum() / mask_anchor_bool.numel()).detach()
    from .entropy_bottleneck import HyperBitSum
    hyper_sum = bit_hyper = None
    if isinstance(likelihood_hyper, HyperBitSum):       # the fused training path already summed the bits
        hyper_sum, n_hyper = likelihood_hyper.total, likelihood_hyper.numel
    elif likelihood_hyper.shape[0] != n:        # already restricted to the chosen rows (context_model_coding_order)
        bit_hyper = -torch.log2(likelihood_hyper)
    else:
        bit_hyper = -torch.log2(gather_unique(likelihood_hyper, torch.nonzero(_as_mask(choose_mask))[:, 0]))
    if bit_hyper is not None:
        hyper_sum, n_hyper = torch.sum(bit_hyper).reshape(1), bit_hyper.numel()
    eg = pc.entropy_gaussian
    all_fused = bool(levels) and all(L.get("fused") for L in levels) and pc._anchor_feat.is_cuda
    if all_fused:
        # every level goes through the fused rate kernel, which takes the three clamp centres as constants: one
        # launch over the three parameter tensors (exp of the scaling logits on the fly) instead of exp + 3 reductions
        xm_feat = xm_scaling = xm_offsets = None
        src0 = next((L["side_src"] for L in levels if L.get("side_src") is not None), None)
        # (valid when the level kernels read exactly pc's parameters: multi_scale_generating is also callable on other tensors)
        s_is_pc = src0 is not None and ((pc.decoded_version and src0.s_orig is pc._scaling)
                                        or getattr(src0.s_orig, "_cgs_exp_of", None) is pc._scaling)
        if (src0 is not None and src0.rows_read == n and src0.f.data_ptr() == pc._anchor_feat.data_ptr()
                and src0.o.data_ptr() == pc._offset.data_ptr() and s_is_pc):
            x_means_fused = src0.means()        # accumulated by the level kernels while they read the rows
        else:
            x_means_fused = _ctx.means3(pc._anchor_feat, pc._scaling, pc._offset, exp_b=not pc.decoded_version)
    else:
        xm_feat, xm_scaling, xm_offsets = pc._anchor_feat.mean(), pc.get_scaling.mean(), pc._offset.mean()
    if (all_fused and not return_sum_bits and all(L.get("chosen") is not None for L in levels)
            and (mask_anchor_bool is None or live_count is not None)):
        # every level on the fused rate kernel, the subset listed level after level, the live fraction a host number:
        # the whole of :1658-1705 is ONE autograd node (a launch per level + one for the scalar tail, each way)
        chosen_all = levels[0]["chosen"][0]
        masks_chosen = (masks_chosen_given if masks_chosen_given is not None
                        else gather_unique(binary_grid_masks.reshape(n, K), chosen_all))
        tensors, spans, sides, maps, level_rows = [], [], [], [], []
        n_feat = n_scaling = n_offsets = 0
        for L in levels:
            n_sub = int(L["loc"].shape[0])
            tensors += [L["yf"], L["ys"], L["yo"], L["Q"], L["pred"], L["loc"]]
            spans.append((L["chosen"][1], L["chosen"][2]))
            sides.append(L.get("side"))
            maps.append(L.get("sub_map"))
            n_feat, n_scaling, n_offsets = n_feat + n_sub * pc.feat_dim, n_scaling + n_sub * 6, n_offsets + n_sub * 3 * K
            level_rows.append(n_sub)
        dead = 0.0 if mask_anchor_bool is None else 1.0 - live_count / mask_anchor_bool.numel()
        meta = dict(use_clamp=_enc.use_clamp, K=K, spans=spans, sides=sides, maps=maps,
                    finish=(float(mask_anchor_rate), float(n_feat), float(n_scaling), float(n_offsets), float(dead)),
                    sum_table=src0.rate_sum_table(len(levels)) if src0 is not None else None)
        out4, raw = _ctx.rate_all(hyper_sum, masks_chosen, x_means_fused, meta, tensors)      # (out4: four 0-dim tensors)
        feat_dim = pc.feat_dim + 6 + 3 * K
        divisors = [1.0, float(max(1, n_hyper))] + [float(max(1, r) * feat_dim) for r in level_rows]
        each_level_bpp = LevelBppReport(raw, [L["n_level"] / n for L in levels], divisors)
        return out4[0], out4[1], out4[2], out4[3], each_level_bpp
    if any(L.get("lazy") for L in levels):
        raise RuntimeError("rate_model: levels with a lazy mean / scale branch need the one-node rate path")
    masks30 = None                              # [N, 3K] mask weights, only the unfused levels read it
    zero = torch.zeros((), device=dev)
    s_feat, s_scaling, s_offsets = zero, zero, zero
    n_feat = n_scaling = n_offsets = 0
    level_bpp_sums, level_rows = [], []
    x_means = masks_chosen = None
    fused_sums = []
    # the fused levels write their three sums into consecutive rows of one zeroed table (rate_finish reads it whole)
    S_table = torch.zeros(len(levels), 3, dtype=torch.float32, device=dev) if all_fused else None
    for L in levels:
        if L.get("fused"):                      # one launch: gathers of the chosen rows + the three rate terms + sums
            if x_means is None:
                x_means = x_means_fused if all_fused else torch.stack([xm_feat, xm_scaling, xm_offsets]).detach()
            n_sub = int(L["loc"].shape[0])
            if L.get("chosen") is not None:
                # the mask rows of ALL levels' chosen anchors in one gather (one scatter + one zero fill on the way
                # back instead of an [N,K] gradient buffer and an accumulation per level)
                if masks_chosen is None:
                    masks_chosen = (masks_chosen_given if masks_chosen_given is not None
                                    else gather_unique(binary_grid_masks.reshape(n, K), L["chosen"][0]))
                m_rows, g_rows = masks_chosen[L["chosen"][1]:L["chosen"][2]], None
            else:
                m_rows, g_rows = binary_grid_masks.reshape(n, K), L["rows"]
            sums = _ctx.level_rate(L["yf"], L["ys"], L["yo"], L["Q"], L["pred"], L["loc"], m_rows, g_rows, x_means,
                                   _enc.use_clamp, K, side=L.get("side"),
                                   out=S_table[len(fused_sums)] if S_table is not None else None)
            fused_sums.append(sums)
            n_feat, n_scaling, n_offsets = n_feat + n_sub * pc.feat_dim, n_scaling + n_sub * 6, n_offsets + n_sub * 3 * K
            level_rows.append(n_sub)
            continue
        if L["selected"]:                       # the level already holds the chosen rows only
            rows = L["rows"]
            g = lambda t: t
        else:
            loc = torch.nonzero(choose_mask[L["orig"]])[:, 0]
            rows = L["orig"][loc]
            g = lambda t, loc=loc: gather_unique(t, loc)
        bf = eg(g(L["feat"]), g(L["mf"]), g(L["sf"]), g(L["qf"]), xm_feat)
        bs = eg(g(L["scaling"]), g(L["ms"]), g(L["ss"]), g(L["qs"]), xm_scaling)
        if masks30 is None:
            masks30 = binary_grid_masks.repeat(1, 1, 3).view(-1, 3 * K)
        bo = eg(g(L["offsets"]), g(L["mo"]), g(L["so"]), g(L["qo"]), xm_offsets) * gather_unique(masks30, rows)
        s_feat, s_scaling, s_offsets = s_feat + bf.sum(), s_scaling + bs.sum(), s_offsets + bo.sum()
        n_feat, n_scaling, n_offsets = n_feat + bf.numel(), n_scaling + bs.numel(), n_offsets + bo.numel()
        level_rows.append(int(rows.shape[0]))
        level_bpp_sums.append((bf.detach().sum() + bs.detach().sum() + bo.detach().sum()))

    if fused_sums and len(fused_sums) == len(levels) and not return_sum_bits:
        feat_dim = pc.feat_dim + 6 + 3 * K
        divisors = [1.0, float(max(1, n_hyper))] + [float(max(1, r) * feat_dim) for r in level_rows]
        if S_table is not None and (mask_anchor_bool is None or live_count is not None):
            # all levels came from the fused rate kernel and the live fraction is a host number: the whole scalar tail
            # of :1687-1705 is one launch (forward) / one launch (backward)
            dead = 0.0 if mask_anchor_bool is None else 1.0 - live_count / mask_anchor_bool.numel()
            out4, raw = _ctx.rate_finish(fused_sums, hyper_sum, mask_anchor_rate, n_feat, n_scaling, n_offsets, dead)
            each_level_bpp = LevelBppReport(raw, [L["n_level"] / n for L in levels], divisors)
            return out4[0], out4[1], out4[2], out4[3], each_level_bpp
        # the same as a handful of [3]-vector ops (the element-by-element version is ~60 one-element launches per step)
        S = torch.stack(fused_sums)                               # [levels, 3] = (feat, scaling, offsets) bits
        tot = S.sum(dim=0) * mask_anchor_rate
        s_hyper = hyper_sum.reshape(()) * mask_anchor_rate
        bit_per_feat_param = tot[0] / max(1, n_feat)
        bit_per_scaling_param = tot[1] / max(1, n_scaling)
        bit_per_offsets_param = tot[2] / max(1, n_offsets)
        bit_per_param = (tot.sum() + s_hyper) / max(1, n_feat + n_scaling + n_offsets)
        with torch.no_grad():
            raw = torch.cat([(1 - mask_anchor_bool.float().mean()).reshape(1) if mask_anchor_bool is not None
                             else torch.zeros(1, device=dev), s_hyper.detach().reshape(1), S.detach().sum(dim=1)])
        each_level_bpp = LevelBppReport(raw, [L["n_level"] / n for L in levels], divisors)
        return bit_per_param, bit_per_feat_param, bit_per_scaling_param, bit_per_offsets_param, each_level_bpp
    for sums in fused_sums:                     # mixed fused / unfused levels: element-wise bookkeeping
        s_feat, s_scaling, s_offsets = s_feat + sums[0], s_scaling + sums[1], s_offsets + sums[2]
        level_bpp_sums.append(sums.detach().sum())

    if return_sum_bits:                                                                # :1672-1685
        bit_anchor = (n_hyper // max(1, pc.latent_codec.channels if hasattr(pc.latent_codec, "channels") else 1)
                      if bit_hyper is None else bit_hyper.shape[0]) * 3 * 16
        bit_masks_sum = get_binary_vxl_size(binary_grid_masks)[1].item()
        return (bit_anchor, hyper_sum.sum().item(), s_feat.item(), s_scaling.item(), s_offsets.item(), bit_masks_sum)

    s_hyper = hyper_sum.reshape(())
    bit_per_hyper_param = s_hyper / max(1, n_hyper) * mask_anchor_rate
    bit_per_feat_param = s_feat / max(1, n_feat) * mask_anchor_rate
    bit_per_scaling_param = s_scaling / max(1, n_scaling) * mask_anchor_rate
    bit_per_offsets_param = s_offsets / max(1, n_offsets) * mask_anchor_rate
    bit_per_param = (s_feat + s_scaling + s_offsets + s_hyper) / max(1, n_feat + n_scaling + n_offsets) * mask_anchor_rate

    # per-level bpp (:1697-1705); ONE host read for the whole report instead of one .item() per level
    with torch.no_grad():
        feat_dim = pc.feat_dim + 6 + 3 * K
        stats = [1 - (mask_anchor_bool.float().mean() if mask_anchor_bool is not None else torch.ones((), device=dev)),
                 bit_per_hyper_param.detach()]
        stats += [s / max(1, r) / feat_dim for s, r in zip(level_bpp_sums, level_rows)]
        dev_stats = torch.stack([t.reshape(()).float() for t in stats])
    each_level_bpp = LevelBppReport(dev_stats, [L["n_level"] / n for L in levels])
    return bit_per_param, bit_per_feat_param, bit_per_scaling_param, bit_per_offsets_param, each_level_bpp


class LevelBppReport(list):
    """`each_level_bpp` of the reference (:1697-1705): [masked-anchor ratio, hyper bpp, [level ratio, level bpp]...].

    The reference fills it with `.item()` reads in the middle of the training step; the numbers are only logged,
    so here the device values stay on the device until the list is first looked at (indexing, iteration, len,
    printing) — one D2H read then, none (and no drained launch queue) for the iterations that do not log."""

    def __init__(self, dev_stats, level_ratios, divisors=None):
        super().__init__()
        self._dev, self._ratios, self._div = dev_stats, level_ratios, divisors

    def _fill(self):
        if self._dev is not None:
            host = self._dev.cpu().tolist()
            if self._div is not None:           # normalisations that only the report needs are done on the host
                host = [v / d for v, d in zip(host, self._div)]
            self._dev = None
            super().extend([host[0], host[1]] + [[r, host[2 + i]] for i, r in enumerate(self._ratios)])
        return self

    def __getitem__(self, k): return list.__getitem__(self._fill(), k)
    def __iter__(self): return list.__iter__(self._fill())
    def __len__(self): return list.__len__(self._fill())
    def __repr__(self): return list.__repr__(self._fill())
    def __eq__(self, other): return list.__eq__(self._fill(), other)
    __hash__ = None


def _unpermute(c, t):
    """coding order -> anchor order; anchors no level codes (none in practice: level 0 takes every leftover)
    read as zeros like the reference's zero-initialised buffers (:1547-1549)."""
    if c["covers_all"]:
        return gather_unique(t, c["inv_perm"])
    out = torch.zeros((c["inv_perm"].shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    return out.index_copy(0, c["perm"], t)


def multi_scale_generating(pc, anchor, hyper, feat, grid_offsets, grid_scaling, binary_grid_masks,
                           mask_anchor_bool=None, training=False, predict_bpp=False, return_sum_bits=False):   # :1541-1707
    # The rate subset is drawn BEFORE the level loop (the reference draws it after, :1659) so that the loop can
    # skip the mean/scale outputs of unchosen anchors; same distribution, different position in the RNG stream.
    choose_mask = None
    if predict_bpp:
        choose_mask = (choose_mask_provider(anchor, mask_anchor_bool) if (choose_mask_provider is not None and not return_sum_bits)
                       else draw_choose_mask(anchor, mask_anchor_bool, return_sum_bits))
    c, feat_p, scal_p, off_p, likelihood_hyper, levels = context_model_coding_order(
        pc, anchor, hyper, feat, grid_offsets, grid_scaling, mask_anchor_bool, training, keep_stats=predict_bpp,
        choose_mask=choose_mask, allow_rate_lazy=not return_sum_bits)
    if predict_bpp and return_sum_bits:
        return rate_model(pc, anchor, binary_grid_masks, mask_anchor_bool, likelihood_hyper, levels, True, choose_mask)
    feat_after_Q, grid_scaling_after_Q, grid_offsets_after_Q = (_unpermute(c, t) for t in (feat_p, scal_p, off_p))
    if not predict_bpp:
        return feat_after_Q, grid_scaling_after_Q, grid_offsets_after_Q
    # (levels whose mean / scale branch lives in the rate kernels need the one-node rate path, i.e. the live count as a host number)
    lazy = any(L.get("lazy") for L in levels)
    rates = rate_model(pc, anchor, binary_grid_masks, mask_anchor_bool, likelihood_hyper, levels, False, choose_mask,
                       live_count=c.get("live_count") if lazy else None)
    return (feat_after_Q, grid_scaling_after_Q, grid_offsets_after_Q) + tuple(rates)


class LazyRows:
    """src[idx] (distinct rows) not yet materialised."""

    def __init__(self, src, idx):
        self.src, self.idx = src, idx

    def materialize(self):
        return gather_unique(self.src, self.idx)

    def cat_with(self, *others):
        """cat([src[idx], *others], dim=1) as one rowcat launch."""
        return _ctx.rowcat([(self.src, self.idx, True)] + [(o, None, True) for o in others])


def multi_scale_generating_visible(pc, anchor, hyper, feat, grid_offsets, grid_scaling, binary_grid_masks,
                                   mask_anchor_bool, vis_idx, training, predict_bpp, defer_feat=False, begun=None,
                                   defer_rate=False):
    """multi_scale_generating followed by `[visible_mask]` (gaussian_renderer/__init__.py:73-81, 93-101) with the
    two row gathers composed into one: out[k] = Q_coding_order[inv_perm[vis_idx[k]]].  defer_feat: return the
    feature rows as a LazyRows (source + row index) so that the caller can fuse the gather into its own kernel."""
    # the rate subset (:1658-1661): drawn inside the step's bookkeeping kernel on the device (same distribution, the
    # build's counter-based generator instead of torch's Philox stream), by torch otherwise
    choose_mask, draw = None, False
    if predict_bpp:
        if begun is not None and begun["given"] is not None:
            choose_mask = begun["given"]
        elif choose_mask_provider is not None:
            choose_mask = choose_mask_provider(anchor, mask_anchor_bool)
        elif anchor.is_cuda:
            draw = True
        else:
            choose_mask = draw_choose_mask(anchor, mask_anchor_bool, False)
    c, feat_p, scal_p, off_p, likelihood_hyper, levels = context_model_coding_order(
        pc, anchor, hyper, feat, grid_offsets, grid_scaling, mask_anchor_bool, training, keep_stats=predict_bpp,
        choose_mask=choose_mask, draw_choose=draw, begun=begun, allow_rate_lazy=True)
    if draw:
        choose_mask = c.get("_choose_mask")
    if c["covers_all"]:
        # (the renderer may have formed the rows already, for its early launch of the anchor MLPs: the same tensor is handed on)
        e_pos = begun.get("early_pos") if begun is not None else None
        pos = e_pos[1] if (e_pos is not None and e_pos[0] is vis_idx and e_pos[2] is c) else _index_rows(c["inv_perm"], vis_idx)
        lazy = defer_feat and feat_p.is_cuda and feat_p.dtype == torch.float32 and LAZY_MODE > 0
        if lazy and LAZY_MODE == 1:
            outs = (LazyRows(feat_p, pos), gather_unique(scal_p, pos), gather_unique(off_p, pos))
        elif lazy:      # the consumers (MLP-input assembly, expansion kernels) gather the rows themselves
            outs = (LazyRows(feat_p, pos), LazyRows(scal_p, pos), LazyRows(off_p, pos))
        else:
            outs = tuple(gather_unique(t, pos) for t in (feat_p, scal_p, off_p))
    else:
        outs = tuple(gather_unique(_unpermute(c, t), vis_idx) for t in (feat_p, scal_p, off_p))
    if not predict_bpp:
        return outs
    # (binary_grid_masks may be a callable: the caller creates the mask's autograd node after the level loop's nodes)
    rate = lambda masks_chosen=None: tuple(rate_model(
        pc, anchor, binary_grid_masks() if callable(binary_grid_masks) else binary_grid_masks, mask_anchor_bool, likelihood_hyper,
        levels, False, choose_mask, live_count=c.get("live_count"), masks_chosen_given=masks_chosen))
    # the rows whose mask weights the rate model will gather, when it is one list for all levels (the caller may gather them in
    # the same node as its own rows of the mask: _GatherUniquePair)
    rate.chosen_rows = (levels[0]["chosen"][0] if (levels and all(L.get("fused") and L.get("chosen") is not None
                                                                     and L["chosen"][0] is levels[0]["chosen"][0] for L in levels))
                        else None)
    if defer_rate:      # the caller enqueues the rate model where it fills a read-back bubble (renderer.py)
        return outs + (rate,)
    return outs + rate()
