"""Shim for the absent `torch_scatter` wheel: the one function the reference imports
(`from torch_scatter import scatter_max`, scene/gaussian_model.py:18; used at :826,829)."""
import torch


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    if dim != 0:
        raise NotImplementedError("shim implements dim=0 (the reference's only use)")
    n = dim_size if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
    res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device) if out is None else out
    res.scatter_reduce_(0, index, src, reduce="amax", include_self=out is not None)
    return res, None      # the reference reads [0] only (argmax is not needed)
