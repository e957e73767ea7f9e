"""Drop-in for the reference's utils/multi_level.py (`torch_unique_with_indices`).

Contract (verified against torch on CPU): rows come back in ascending
lexicographic order (col 0, then 1, then 2; -0.0 merges with 0.0), `inverse` maps every
input row to its unique row, `indices` is the SMALLEST original index of each group
(the reference's float64 scatter_reduce amin, utils/multi_level.py:23-30), `counts` the
group sizes.

Integer-valued voxel keys (the only inputs on the hot path: torch.round(anchor/voxel/
scale), scene/gaussian_model.py:1760) run on the device (csrc/levels.hip, round 3): range /
integrality reduction, the three coordinates biased and packed into one <= 62-bit key,
libcgs_hip's stable LSD radix sort of (key, index) (one sort, or two for keys wider than 32
bits), run heads, scan, and one kernel that writes inverse / first occurrence / unique rows —
ascending packed key == ascending lexicographic row order, stability gives the first-
occurrence index for free.  Anything else (non-integer rows, huge ranges, other dims) is not
on the hot path and raises.
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
    import ctypes as C
    if dim != 0 or tensor.dim() != 2:
        raise NotImplementedError("hot path only needs dim=0 on [N,C] voxel keys")
    _lib.require_device(tensor)
    with torch.no_grad():
        n, c = tensor.shape
        dev = tensor.device
        if n == 0:
            e = torch.zeros(0, dtype=torch.long, device=dev)
            return tensor.clone(), e, e.clone(), e.clone()
        if c != 3 or tensor.dtype != torch.float32:
            raise NotImplementedError("torch_unique_with_indices: [N,3] fp32 voxel keys only")
        L = _lib.lib()
        t = tensor.contiguous()
        stream = _lib.current_stream()
        # launch 1-3: per-column range + integrality; ONE host read decides the key widths
        out7 = torch.empty(7, dtype=torch.float32, device=dev)
        tmp8 = torch.empty(8, dtype=torch.int32, device=dev)
        _lib.check(L.cgs_level_key_range(_lib.ptr(t), n, _lib.ptr(out7), _lib.ptr(tmp8), stream), "cgs_level_key_range")
        host = out7.tolist()
        if host[6] != 0.0:
            raise NotImplementedError("torch_unique_with_indices: rows must be integer-valued voxel keys")
        lo_l = [int(v) for v in host[:3]]
        bits = [max(1, int(int(hi) - l).bit_length()) for hi, l in zip(host[3:6], lo_l)]
        if sum(bits) > 62 or max(bits) > 31:
            raise NotImplementedError("voxel key range too large to pack")
        inverse = torch.empty(n, dtype=torch.long, device=dev)
        first = torch.empty(n, dtype=torch.long, device=dev)
        counts = torch.empty(n, dtype=torch.long, device=dev)
        unique = torch.empty(n, 3, dtype=torch.float32, device=dev)
        scratch = torch.empty(int(L.cgs_level_unique_scratch_bytes(n)), dtype=torch.uint8, device=dev)
        m = C.c_int64(0)
        _lib.check(L.cgs_level_unique(_lib.ptr(t), n, (C.c_int32 * 3)(*lo_l), (C.c_int32 * 3)(*bits), _lib.ptr(inverse),
                                      _lib.ptr(first), _lib.ptr(counts), _lib.ptr(unique), C.byref(m), _lib.ptr(scratch),
                                      scratch.numel(), stream), "cgs_level_unique")
        k = int(m.value)
        return unique[:k], inverse, first[:k], counts[:k]
