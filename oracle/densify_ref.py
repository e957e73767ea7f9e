"""TEST INFRASTRUCTURE ONLY — numpy restatement of the reference's densification statistics and anchor growing,
the checker for contextgs_amd/densify.py and csrc/densify.hip.  Imported only by tests/.

Follows scene/gaussian_model.py of the reference:
  training_statis  :696-713
  anchor_growing   :762-855   (candidate selection, voxel de-duplication against the existing anchors,
                               per-voxel feature max, new-anchor attributes; including the quirk that rounds
                               i > 0 are skipped while no anchor has been added, :774-777)
  Quantize_anchor  utils/encodings.py:219-231 (get_anchor)
Pinned by tests/golden/densify.npz (outputs of the reference's own methods, tools/make_densify_golden.py).
"""
import numpy as np

f32 = np.float32


def training_statis(K, vis, opacity, sel, update_filter, grad, opacity_accum, anchor_demon, grad_accum, denom):
    """Returns the four accumulators after one call (inputs are not modified)."""
    oa, ad, ga, dn = (a.copy() for a in (opacity_accum, anchor_demon, grad_accum, denom))
    t = np.maximum(opacity.reshape(-1), f32(0)).reshape(-1, K)
    oa[vis] += t.sum(axis=1, keepdims=True, dtype=f32)
    ad[vis] += f32(1)
    slot_vis = np.repeat(vis, K)
    combined = np.zeros(ga.shape[0], dtype=bool)
    combined[slot_vis] = sel
    idx = np.nonzero(combined)[0][update_filter]
    g = grad[update_filter, :2]
    ga[idx, 0] += np.sqrt((g * g).sum(axis=1, dtype=f32))
    dn[idx, 0] += f32(1)
    return oa, ad, ga, dn


def quantize_anchor(anchor, lo, hi):
    interval = (hi - lo) / f32(65535.0) + f32(1e-6)
    q = np.clip(np.floor((anchor - lo) / interval), 0, 65535).astype(f32)
    return q * interval + lo


def anchor_growing(anchor, offset, scaling, feat, hyper, lo, hi, grads, threshold, offset_mask, rands, voxel_size,
                   K, update_depth=3, init_factor=100, hier=4):
    """Returns the list of per-round dicts the reference hands to cat_tensors_to_optimizer (rounds that add nothing
    are absent) — keys anchor, scaling, anchor_feat, hyper_latent, depth."""
    anchor, offset, scaling, feat, hyper = (a.copy() for a in (anchor, offset, scaling, feat, hyper))
    init_length = anchor.shape[0] * K
    rounds = []
    for i in range(update_depth):
        cur_threshold = threshold * ((hier // 2) ** i)
        cand = (grads >= f32(cur_threshold)) & offset_mask
        cand &= rands[i] > f32(0.5 ** (i + 1))
        length_inc = anchor.shape[0] * K - init_length
        if length_inc == 0:
            if i > 0:
                continue
        else:
            cand = np.concatenate([cand, np.zeros(length_inc, dtype=bool)])
        qa = quantize_anchor(anchor, lo, hi)
        all_xyz = qa[:, None, :] + offset * np.exp(scaling[:, :3])[:, None, :]
        size_factor = init_factor // (hier ** i)
        cur_size = f32(voxel_size * size_factor)
        grid = np.rint(qa / cur_size).astype(np.int32)
        sel_xyz = all_xyz.reshape(-1, 3)[cand]
        sel_grid = np.rint(sel_xyz / cur_size).astype(np.int32)
        uniq, inverse = np.unique(sel_grid, axis=0, return_inverse=True)
        inverse = inverse.reshape(-1)
        existing = {tuple(r) for r in grid.tolist()}
        keep = np.array([tuple(r) not in existing for r in uniq.tolist()], dtype=bool)
        cand_anchor = uniq[keep].astype(f32) * cur_size
        if cand_anchor.shape[0] == 0:
            continue
        M = cand_anchor.shape[0]

        def group_max(src):
            out = np.full((uniq.shape[0], src.shape[1]), -np.inf, dtype=f32)
            np.maximum.at(out, inverse, src)
            return out[keep]

        new_feat = group_max(np.repeat(feat, K, axis=0)[cand])
        new_hyper = group_max(np.repeat(hyper, K, axis=0)[cand])
        new_scaling = np.log(np.ones((M, 6), dtype=f32) * cur_size)
        rounds.append(dict(anchor=cand_anchor, scaling=new_scaling, anchor_feat=new_feat, hyper_latent=new_hyper, depth=i))
        anchor = np.concatenate([anchor, cand_anchor])
        offset = np.concatenate([offset, np.zeros((M, K, 3), dtype=f32)])
        scaling = np.concatenate([scaling, new_scaling])
        feat = np.concatenate([feat, new_feat])
        hyper = np.concatenate([hyper, new_hyper])
    return rounds
