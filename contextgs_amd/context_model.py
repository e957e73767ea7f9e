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
    indices to code and the level anchors of those rows."""
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
        plan.append((i, to_code, orig, hybrid_anchor_list[i][to_code]))
    return plan, inverse_indices_list, mapping_list


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


def multi_scale_generating(pc, anchor, hyper, feat, grid_offsets, grid_scaling, binary_grid_masks,
                           mask_anchor_bool=None, training=False, predict_bpp=False, return_sum_bits=False):   # :1541-1707
    K = pc.n_offsets
    n = anchor.shape[0]
    dev = anchor.device
    content_pre_gathered = None
    to_code_list = []

    feat_after_Q = torch.zeros_like(feat)
    grid_scaling_after_Q = torch.zeros_like(grid_scaling)
    grid_offsets_after_Q = torch.zeros_like(grid_offsets)
    already_coded = torch.zeros(n, dtype=torch.bool, device=dev)
    if predict_bpp:
        z = torch.zeros_like
        off2 = grid_offsets.reshape(-1, 3 * K)
        mean_feat_all, scale_feat_all, Q_feat_all = z(feat), z(feat), z(feat[:, [0]])
        mean_scaling_all, scale_scaling_all, Q_scaling_all = z(grid_scaling), z(grid_scaling), z(grid_scaling[:, [0]])
        mean_offsets_all, scale_offsets_all, Q_offsets_all = z(off2), z(off2), z(off2[:, [0]])

    hyper_feat, likelihood_hyper = pc.latent_codec(hyper, training=training)           # :1556
    if pc.disable_hyper:
        hyper_feat = hyper_feat * 0
    if pc.level_scale is None:                                                         # :1559
        sel = anchor[mask_anchor_bool] if mask_anchor_bool is not None else anchor
        pc.level_scale = find_divide_scale(pc, sel, pc.target_ratio, pc.level_num)
    plan, inverse_indices_list, mapping_list = level_plan(pc, anchor, mask_anchor_bool)

    for (i, to_code, orig, hybrid_anchor) in plan:
        if int(orig.shape[0]) > 0:
            hybrid_feat = feat[orig]
            hybrid_grid_scaling = grid_scaling[orig]
            hybrid_grid_offsets = grid_offsets[orig]
            to_code_list.append(orig)
            if content_pre_gathered is None:                                           # :1596-1600
                feat_in = torch.cat([hybrid_anchor, hyper_feat[orig].float()], dim=1)
            else:
                feat_in = torch.cat([content_pre_gathered, hyper_feat[orig]], dim=1)
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

            if predict_bpp:                                                            # :1629-1641
                mean_feat_all[orig] = mean_feat; scale_feat_all[orig] = scale_feat; Q_feat_all[orig] = Q_feat
                mean_scaling_all[orig] = mean_scaling; scale_scaling_all[orig] = scale_scaling; Q_scaling_all[orig] = Q_scaling
                mean_offsets_all[orig] = mean_offsets; scale_offsets_all[orig] = scale_offsets; Q_offsets_all[orig] = Q_offsets

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
    sel = torch.nonzero(choose_mask)[:, 0]
    bit_hyper = -torch.log2(likelihood_hyper[sel])
    eg = pc.entropy_gaussian
    bit_feat = eg(feat_after_Q[sel], mean_feat_all[sel], scale_feat_all[sel], Q_feat_all[sel], pc._anchor_feat.mean())
    bit_scaling = eg(grid_scaling_after_Q[sel], mean_scaling_all[sel], scale_scaling_all[sel], Q_scaling_all[sel],
                     pc.get_scaling.mean())
    bit_offsets = eg(grid_offsets_after_Q[sel].view(-1, 3 * K), mean_offsets_all[sel], scale_offsets_all[sel],
                     Q_offsets_all[sel], pc._offset.mean())
    bit_offsets = bit_offsets * binary_grid_masks[sel].repeat(1, 1, 3).view(-1, 3 * K)

    if return_sum_bits:                                                                # :1672-1685
        bit_anchor = bit_hyper.shape[0] * 3 * 16
        bit_masks_sum = get_binary_vxl_size(binary_grid_masks)[1].item()
        return (bit_anchor, torch.sum(bit_hyper).item(), torch.sum(bit_feat).item(), torch.sum(bit_scaling).item(),
                torch.sum(bit_offsets).item(), bit_masks_sum)

    s_feat, s_scaling, s_offsets, s_hyper = torch.sum(bit_feat), torch.sum(bit_scaling), torch.sum(bit_offsets), torch.sum(bit_hyper)
    bit_per_hyper_param = s_hyper / bit_hyper.numel() * mask_anchor_rate
    bit_per_feat_param = s_feat / bit_feat.numel() * mask_anchor_rate
    bit_per_scaling_param = s_scaling / bit_scaling.numel() * mask_anchor_rate
    bit_per_offsets_param = s_offsets / bit_offsets.numel() * mask_anchor_rate
    bit_per_param = (s_feat + s_scaling + s_offsets + s_hyper) / \
                    (bit_feat.numel() + bit_scaling.numel() + bit_offsets.numel()) * mask_anchor_rate

    # per-level bpp (:1697-1705); one host read for the whole report instead of one .item() per level
    with torch.no_grad():
        bpp_sum_map = bit_offsets.sum(dim=1) + bit_scaling.sum(dim=1) + bit_feat.sum(dim=1)
        feat_dim = pc.feat_dim + 6 + 3 * K
        level_of = torch.full((n,), -1, dtype=torch.long, device=dev)
        for li, index in enumerate(to_code_list):
            level_of[index] = li
        lv = level_of[sel]
        stats = [1 - (mask_anchor_bool.float().mean() if mask_anchor_bool is not None else torch.ones((), device=dev)),
                 bit_per_hyper_param.detach() if torch.is_tensor(bit_per_hyper_param) else torch.tensor(float(bit_per_hyper_param), device=dev)]
        for li in range(len(to_code_list)):
            stats.append(bpp_sum_map[lv == li].mean() / feat_dim)
        host = torch.stack([s.reshape(()).float() for s in stats]).cpu().tolist()
    each_level_bpp = [host[0], host[1]]
    for li, index in enumerate(to_code_list):
        each_level_bpp.append([index.shape[0] / n, host[2 + li]])

    return (feat_after_Q, grid_scaling_after_Q, grid_offsets_after_Q, bit_per_param, bit_per_feat_param,
            bit_per_scaling_param, bit_per_offsets_param, each_level_bpp)
