"""Anchor-level hierarchical context model: drop-in for
the module-level functions of the reference's scene/gaussian_model.py:1541-1792
(`multi_scale_generating`, `extract_context_feat`, `find_divide_scale`,
`divide_levels`, `mapping_to_orign`, `index_of_level_L_in_orign`) and
utils/multi_level.py (`torch_unique_with_indices`).

Same level semantics including the reference's quirks (level-1
context rows are paired in lexicographic-voxel order vs ascending original
index; Q4: the training variant zeroes masked anchors before level 1; Q5:
fp32 left-to-right voxel key).  Quantisation and rate run in the fused HIP
kernels of elementwise.hip (encodings.STE_multistep, entropy_models);
grid MLPs go through rocBLAS (north_star).

Cited lines are scene/gaussian_model.py unless stated otherwise.
"""
from __future__ import annotations

import torch

from . import mlp as _mlp
from .encodings import STE_multistep, get_binary_vxl_size
from .multi_level import torch_unique_with_indices

Q_FEAT0, Q_SCALING0, Q_OFFSETS0 = 1, 0.001, 0.2      # :1564-1566


def mapping_to_orign(mapping_list, L, mask=None):                       # :1768-1787
    assert L > 0, "If L=0, the orgin space can be directly obtained"
    level = L - 1
    mapping_prev = mapping_list[level] if mask is None else mapping_list[level][mask]
    for i in reversed(range(level)):
        mapping_prev = mapping_list[i][mapping_prev]
    return mapping_prev


def index_of_level_L_in_orign(mapping_list, inverse_indices_list, to_be_gathered_index, L):   # :1789-1792
    tmp = to_be_gathered_index
    for i in range(L):
        tmp = inverse_indices_list[i][tmp]
    for i in reversed(range(L)):
        tmp = mapping_list[i][tmp]
    return tmp


def find_divide_scale(pc, anchor, target_ratio, level_num):             # :1726-1749
    scale_upper = ((pc.x_bound_max - pc.x_bound_min) / pc.voxel_size).max()

    def binary_search(scale_upper, scale_lower, anchor, target_ratio):
        while True:
            scale = (scale_upper + scale_lower) / 2
            anchor_unique = torch_unique_with_indices(torch.round(anchor / pc.voxel_size / scale), dim=0)[0] * pc.voxel_size * scale
            ratio = anchor_unique.shape[0] / anchor.shape[0]
            if abs(ratio - target_ratio) < 0.01 or (scale_upper - scale_lower).abs() < 1:
                break
            if ratio < target_ratio:
                scale_upper = scale
            else:
                scale_lower = scale
        return scale, anchor_unique

    anchor_unique = anchor
    scale_list = []
    scale_lower = 1
    for _ in range(level_num - 1):
        scale, anchor_unique = binary_search(scale_upper, scale_lower, anchor_unique, target_ratio)
        scale_lower = scale
        scale_list.append(scale.item())
    return scale_list


def divide_levels(pc, anchor, mask_anchor_bool=None):                   # :1751-1765
    hybrid_anchor_list = [anchor]
    inverse_indices_list, mapping_list = [], []
    hybrid_anchor = anchor
    for i in range(1, pc.level_num):
        if i == 1 and mask_anchor_bool is not None:
            hybrid_anchor = hybrid_anchor * mask_anchor_bool.unsqueeze(1)
        _u, inverse_indices, mapping, _c = torch_unique_with_indices(
            torch.round(hybrid_anchor / pc.voxel_size / pc.level_scale[i - 1]), dim=0)
        hybrid_anchor = hybrid_anchor[mapping]
        hybrid_anchor_list.append(hybrid_anchor)
        inverse_indices_list.append(inverse_indices)
        mapping_list.append(mapping)
    return hybrid_anchor_list, inverse_indices_list, mapping_list, hybrid_anchor


def extract_context_feat(anchor_after_Q, feat_after_Q, grid_scaling_after_Q, already_coded, inverse_indices_list,
                         mapping_list, i):                              # :1711-1724
    n = anchor_after_Q.shape[0]
    if i > 1:
        to_be_gathered_mask = torch.zeros(n, dtype=torch.bool, device=anchor_after_Q.device)
        to_be_gathered_mask[mapping_to_orign(mapping_list, i - 1)] = True
    else:
        to_be_gathered_mask = torch.ones(n, dtype=torch.bool, device=anchor_after_Q.device)
    to_be_gathered_mask = to_be_gathered_mask & (~already_coded)
    idx = index_of_level_L_in_orign(mapping_list, inverse_indices_list, torch.nonzero(to_be_gathered_mask)[:, 0], i)
    # gather three row blocks instead of materialising the [N,59] concat (:1712) every level
    return torch.cat([anchor_after_Q[idx], feat_after_Q[idx], grid_scaling_after_Q[idx]], dim=1)


def level_plan(pc, anchor, mask_anchor_bool):
    """Index bookkeeping of the level loop (:1559-1593), shared by the rate model, the
    encoder and the decoder.  Returns per level (from L-1 down to 0) the original-space
    indices to code and the level anchors of those rows.

    The division depends only on (anchor, mask, voxel_size, level_scale); anchors are frozen
    after densification (position lr = 0, arguments/__init__.py:86-87) and the anchor mask
    changes rarely, so the plan of the previous call is reused when both tensors compare equal
    (two elementwise compares + one host read instead of two device sorts and ~40 index
    kernels; SURVEY §8a "caching opportunity")."""
    key = (float(pc.voxel_size), tuple(float(v) for v in pc.level_scale), int(pc.level_num),
           tuple(anchor.shape), mask_anchor_bool is None)
    cache = getattr(pc, "_level_cache", None)
    if cache is not None and cache["key"] == key:
        same = (anchor == cache["anchor"]).all()
        if mask_anchor_bool is not None:
            same = same & (mask_anchor_bool == cache["mask"]).all()
        if bool(same):
            return cache["plan"], cache["inverse"], cache["mapping"]
    plan, inverse_indices_list, mapping_list = _level_plan_uncached(pc, anchor, mask_anchor_bool)
    try:
        pc._level_cache = dict(key=key, anchor=anchor.detach().clone(),
                               mask=None if mask_anchor_bool is None else mask_anchor_bool.clone(), plan=plan,
                               inverse=inverse_indices_list, mapping=mapping_list)
    except Exception:
        pass
    return plan, inverse_indices_list, mapping_list


def _level_plan_uncached(pc, anchor, mask_anchor_bool):
    hybrid_anchor_list, inverse_indices_list, mapping_list, _ = divide_levels(pc, anchor, mask_anchor_bool)
    n = anchor.shape[0]
    dev = anchor.device
    plan = []
    for i in reversed(range(pc.level_num)):
        n_level = n if i == 0 else mapping_list[i - 1].shape[0]
        if i != pc.level_num - 1:
            to_code = torch.ones(n_level, dtype=torch.bool, device=dev)
            to_code[mapping_list[i]] = False
        else:
            to_code = torch.ones(n_level, dtype=torch.bool, device=dev)
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
    return anchor[orig]


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


class _GatherUnique(torch.autograd.Function):
    """x[idx] for UNIQUE row indices: the backward is a plain row scatter into zeros (index_copy_) instead of
    torch's index_put_(accumulate=True), which sorts the indices first (rocprof: ~44 ms of merge-sort +
    indexing_backward kernels per 14 steps at 1 M anchors)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        out = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        out.index_copy_(0, idx, g.contiguous())
        return out, None


def gather_unique(x, idx):
    return _GatherUnique.apply(x, idx) if x.requires_grad else x.index_select(0, idx)


def multi_scale_generating(pc, anchor, hyper, feat, grid_offsets, grid_scaling, binary_grid_masks,
                           mask_anchor_bool=None, training=False, predict_bpp=False, return_sum_bits=False):   # :1541-1707
    K = pc.n_offsets
    n = anchor.shape[0]
    dev = anchor.device
    content_pre_gathered = None
    levels = []          # per coded level: everything the rate model needs, in the level's own row order

    feat_after_Q = torch.zeros_like(feat)
    grid_scaling_after_Q = torch.zeros_like(grid_scaling)
    grid_offsets_after_Q = torch.zeros_like(grid_offsets)
    already_coded = torch.zeros(n, dtype=torch.bool, device=dev)

    hyper_feat, likelihood_hyper = pc.latent_codec(hyper, training=training)           # :1556
    if pc.disable_hyper:
        hyper_feat = hyper_feat * 0
    if pc.level_scale is None:                                                         # :1559
        sel = anchor[mask_anchor_bool] if mask_anchor_bool is not None else anchor
        pc.level_scale = find_divide_scale(pc, sel, pc.target_ratio, pc.level_num)
    plan, inverse_indices_list, mapping_list = level_plan(pc, anchor, mask_anchor_bool)

    for (i, to_code, orig, _unused) in plan:
        if int(orig.shape[0]) > 0:
            # rows of a level are distinct anchors -> sort-free gathers (reference: feat[mapping][to_code], :1569-1585)
            hybrid_feat = gather_unique(feat, orig)
            hybrid_grid_scaling = gather_unique(grid_scaling, orig)
            hybrid_grid_offsets = gather_unique(grid_offsets, orig)
            hyper_l = gather_unique(hyper_feat, orig)
            if content_pre_gathered is None:                                           # :1596-1600
                feat_in = torch.cat([level_anchors(anchor, mask_anchor_bool, i, orig), hyper_l.float()], dim=1)
            else:
                feat_in = torch.cat([content_pre_gathered, hyper_l], dim=1)
            (mean_feat, scale_feat, mean_scaling, scale_scaling, mean_offsets, scale_offsets, Q_feat, Q_scaling,
             Q_offsets) = split_prediction(pc, grid_mlp(pc, i, feat_in))

            if training:                                                               # :1610-1616
                hybrid_feat = hybrid_feat + torch.empty_like(hybrid_feat).uniform_(-0.5, 0.5) * Q_feat
                hybrid_grid_scaling = hybrid_grid_scaling + torch.empty_like(hybrid_grid_scaling).uniform_(-0.5, 0.5) * Q_scaling
                qo = Q_offsets.view(hybrid_feat.shape[0], K, -1) if pc.adaptQ_per_channel else Q_offsets.unsqueeze(1)
                hybrid_grid_offsets = hybrid_grid_offsets + torch.empty_like(hybrid_grid_offsets).uniform_(-0.5, 0.5) * qo
            else:                                                                      # :1617-1625
                hybrid_feat = STE_multistep.apply(hybrid_feat, Q_feat).detach()
                hybrid_grid_scaling = STE_multistep.apply(hybrid_grid_scaling, Q_scaling).detach()
                qo = Q_offsets.view(hybrid_feat.shape[0], K, -1) if pc.adaptQ_per_channel else Q_offsets.unsqueeze(1)
                hybrid_grid_offsets = STE_multistep.apply(hybrid_grid_offsets, qo).detach()
            hybrid_grid_offsets = hybrid_grid_offsets.reshape(-1, 3 * K)

            if predict_bpp:
                # The reference scatters these nine tensors into N-sized buffers (:1629-1641) and gathers the
                # 15 % rate subset back out (:1666-1669); the subset of a level is gathered from the level's own
                # rows instead — same elements, no N-sized round trip.
                levels.append(dict(orig=orig, feat=hybrid_feat, scaling=hybrid_grid_scaling, offsets=hybrid_grid_offsets,
                                   mf=mean_feat, sf=scale_feat, qf=Q_feat, ms=mean_scaling, ss=scale_scaling, qs=Q_scaling,
                                   mo=mean_offsets, so=scale_offsets, qo=Q_offsets))
            feat_after_Q[orig] = hybrid_feat                                           # :1644-1647
            grid_scaling_after_Q[orig] = hybrid_grid_scaling
            grid_offsets_after_Q[orig] = hybrid_grid_offsets.view(-1, K, 3)
            already_coded[orig] = True
        if i != 0:                                                                     # :1650-1651
            content_pre_gathered = extract_context_feat(anchor, feat_after_Q, grid_scaling_after_Q, already_coded,
                                                        inverse_indices_list, mapping_list, i)

    if not predict_bpp:
        return feat_after_Q, grid_scaling_after_Q, grid_offsets_after_Q

    # ---- rate (:1657-1707) ----------------------------------------------------------------
    thresh = 1 if return_sum_bits else 0.15
    choose_mask = torch.rand_like(anchor[:, 0]) <= thresh
    if mask_anchor_bool is not None:
        choose_mask = choose_mask & mask_anchor_bool
        mask_anchor_rate = (mask_anchor_bool.sum() / mask_anchor_bool.numel()).detach()
    else:
        mask_anchor_rate = 1
    bit_hyper = -torch.log2(likelihood_hyper[torch.nonzero(choose_mask)[:, 0]])
    eg = pc.entropy_gaussian
    xm_feat, xm_scaling, xm_offsets = pc._anchor_feat.mean(), pc.get_scaling.mean(), pc._offset.mean()
    masks30 = binary_grid_masks.repeat(1, 1, 3).view(-1, 3 * K)
    zero = torch.zeros((), device=dev)
    s_feat, s_scaling, s_offsets = zero, zero, zero
    n_feat = n_scaling = n_offsets = 0
    level_bpp_sums, level_rows = [], []
    for L in levels:
        loc = torch.nonzero(choose_mask[L["orig"]])[:, 0]
        rows = L["orig"][loc]
        g = lambda t: gather_unique(t, loc)
        bf = eg(g(L["feat"]), g(L["mf"]), g(L["sf"]), g(L["qf"]), xm_feat)
        bs = eg(g(L["scaling"]), g(L["ms"]), g(L["ss"]), g(L["qs"]), xm_scaling)
        bo = eg(g(L["offsets"]), g(L["mo"]), g(L["so"]), g(L["qo"]), xm_offsets) * masks30[rows]
        s_feat, s_scaling, s_offsets = s_feat + bf.sum(), s_scaling + bs.sum(), s_offsets + bo.sum()
        n_feat, n_scaling, n_offsets = n_feat + bf.numel(), n_scaling + bs.numel(), n_offsets + bo.numel()
        level_rows.append(int(loc.shape[0]))
        level_bpp_sums.append((bf.detach().sum() + bs.detach().sum() + bo.detach().sum()))

    if return_sum_bits:                                                                # :1672-1685
        bit_anchor = bit_hyper.shape[0] * 3 * 16
        bit_masks_sum = get_binary_vxl_size(binary_grid_masks)[1].item()
        return (bit_anchor, torch.sum(bit_hyper).item(), s_feat.item(), s_scaling.item(), s_offsets.item(), bit_masks_sum)

    s_hyper = torch.sum(bit_hyper)
    bit_per_hyper_param = s_hyper / max(1, bit_hyper.numel()) * mask_anchor_rate
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
        host = torch.stack([t.reshape(()).float() for t in stats]).cpu().tolist()
    each_level_bpp = [host[0], host[1]]
    for li, L in enumerate(levels):
        each_level_bpp.append([L["orig"].shape[0] / n, host[2 + li]])

    return (feat_after_Q, grid_scaling_after_Q, grid_offsets_after_Q, bit_per_param, bit_per_feat_param,
            bit_per_scaling_param, bit_per_offsets_param, each_level_bpp)
