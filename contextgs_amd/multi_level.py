"""Drop-in for the reference's utils/multi_level.py (`torch_unique_with_indices`).

Contract (verified against torch on CPU): rows come back in ascending
lexicographic order (col 0, then 1, then 2; -0.0 merges with 0.0), `inverse` maps every
input row to its unique row, `indices` is the SMALLEST original index of each group
(the reference's float64 scatter_reduce amin, utils/multi_level.py:23-30), `counts` the
group sizes.

Integer-valued voxel keys (the only inputs on the hot path: torch.round(anchor/voxel/
scale), scene/gaussian_model.py:1760) take the HIP route: the three coordinates are
biased and packed into one 63-bit key and sorted with libcgs_hip's stable LSD radix
sort (prims.hip) — ascending packed key == ascending lexicographic row order, stability
gives the first-occurrence index for free.  Anything else (non-integer rows, huge
ranges, other dims) is not on the hot path and raises.
"""
from __future__ import annotations

import torch

from . import _lib


def _sort_pairs_u32(keys: torch.Tensor, vals: torch.Tensor, bits: int):
    L = _lib.lib()
    n = keys.numel()
    ko, vo, kt, vt = (torch.empty_like(keys) for _ in range(4))
    scratch = torch.empty(L.cgs_sort_scratch_bytes(n), dtype=torch.uint8, device=keys.device)
    _lib.check(L.cgs_sort_pairs_u32(_lib.ptr(keys), _lib.ptr(vals), _lib.ptr(ko), _lib.ptr(vo), _lib.ptr(kt),
                                    _lib.ptr(vt), n, 0, bits, _lib.ptr(scratch), scratch.numel(),
                                    _lib.current_stream()), "cgs_sort_pairs_u32")
    return ko, vo


def torch_unique_with_indices(tensor, dim=0):
    """Return (unique rows, inverse_indices, first-occurrence indices, counts)."""
    if dim != 0 or tensor.dim() != 2:
        raise NotImplementedError("hot path only needs dim=0 on [N,C] voxel keys")
    _lib.require_device(tensor)
    with torch.no_grad():
        n, c = tensor.shape
        dev = tensor.device
        if n == 0:
            e = torch.zeros(0, dtype=torch.long, device=dev)
            return tensor.clone(), e, e.clone(), e.clone()
        t = tensor + 0.0                               # -0.0 -> +0.0 so that equal rows get equal keys
        # one host read for (min, max, integrality) — fp32 reductions, not int64 ones (those run ~10x slower)
        not_int = (torch.round(t) != t).any().to(t.dtype).reshape(1)
        host = torch.cat([t.amin(dim=0), t.amax(dim=0), not_int]).cpu().tolist()
        if host[-1] != 0.0:
            raise NotImplementedError("torch_unique_with_indices: rows must be integer-valued voxel keys")
        lo_l = [int(v) for v in host[:c]]
        span = [int(hi) - l + 1 for hi, l in zip(host[c:2 * c], lo_l)]
        bits = [max(1, int(s_ - 1).bit_length()) for s_ in span]
        if sum(bits) > 62 or max(bits) > 31:
            raise NotImplementedError("voxel key range too large to pack")
        lo = torch.tensor(lo_l, dtype=t.dtype, device=dev)
        ti = (t - lo).to(torch.int32)
        rel = ti
        # LSD over columns, least significant (last column) first; each column in <=32-bit digits
        order = torch.arange(n, dtype=torch.int32, device=dev)
        for col in reversed(range(c)):
            keys = rel[:, col][order.long()].contiguous()
            _, order = _sort_pairs_u32(keys, order.contiguous(), bits[col])
        order = order.long()
        sorted_rows = rel[order]
        new_group = torch.ones(n, dtype=torch.bool, device=dev)
        new_group[1:] = (sorted_rows[1:] != sorted_rows[:-1]).any(dim=1)
        gid_sorted = torch.cumsum(new_group.to(torch.int64), 0) - 1
        starts = torch.nonzero(new_group)[:, 0]
        inverse = torch.empty(n, dtype=torch.long, device=dev)
        inverse[order] = gid_sorted
        indices = order[starts]                        # stable sort => first element of a group is its smallest index
        counts = torch.diff(torch.cat([starts, torch.tensor([n], device=dev)]))
        unique = t[indices]
        return unique, inverse, indices, counts
