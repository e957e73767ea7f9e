"""codec_driver._fast_checkpoint_load (the decoder's reader of mlp.pt / meta.b: archive mapped once, tensors as views)
against torch.load on the structures the container holds, and the fall-back for anything it does not resolve.  CPU only."""
import collections
import fractions

import numpy as np
import torch

from contextgs_amd import codec_driver as cd


def _same(x, y):
    if isinstance(x, torch.Tensor):
        return isinstance(y, torch.Tensor) and x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y)
    if isinstance(x, np.ndarray):
        return isinstance(y, np.ndarray) and x.dtype == y.dtype and np.array_equal(x, y)
    if isinstance(x, dict):
        return type(x) is type(y) and list(x) == list(y) and all(_same(x[k], y[k]) for k in x)
    if isinstance(x, (list, tuple)):
        return type(x) is type(y) and len(x) == len(y) and all(_same(a, b) for a, b in zip(x, y))
    return x == y


def _checkpoint():
    g = torch.Generator().manual_seed(3)
    sd = collections.OrderedDict()
    sd["0.weight"] = torch.randn(100, 71, generator=g)
    sd["0.bias"] = torch.randn(100, generator=g)
    sd["t"] = torch.randn(5, 7, generator=g).t()                 # non-contiguous view
    sd["slice"] = torch.arange(40, dtype=torch.int32)[8:20]      # storage offset
    sd["empty"] = torch.zeros(0, 3)
    sd["flag"] = torch.tensor([True, False])
    sd["scalar"] = torch.tensor(2.5, dtype=torch.float64)
    sd["param"] = torch.nn.Parameter(torch.randn(3, generator=g))
    return {"grid_mlp": sd, "bound": [torch.zeros(1, 3), torch.ones(1, 3)], "level_scale": [1.0, 2.5, None],
            "meta": [7, 1000, {0: np.arange(12, dtype=np.int32), 1: np.zeros(0, dtype=np.int64)}, 0.25, ("a", 3)]}


def test_fast_loader_equals_torch_load(tmp_path):
    p = str(tmp_path / "mlp.pt")
    torch.save(_checkpoint(), p)
    fast = cd._fast_checkpoint_load(p)
    ref = torch.load(p, map_location="cpu", weights_only=False)
    ref["grid_mlp"]["param"] = ref["grid_mlp"]["param"].detach()           # the loader hands back the data of a Parameter
    assert _same(fast, ref)
    assert _same(cd.read_mlp_checkpoint(p), ref)
    fast["grid_mlp"]["0.bias"].add_(1.0)                                    # private pages: the file is untouched
    assert _same(cd._fast_checkpoint_load(p)["grid_mlp"]["0.bias"], ref["grid_mlp"]["0.bias"])


def test_unresolved_global_falls_back_to_torch_load(tmp_path):
    p = str(tmp_path / "odd.pt")
    torch.save({"x": torch.arange(3), "q": fractions.Fraction(1, 3)}, p)
    try:
        cd._fast_checkpoint_load(p)
        raise AssertionError("the restricted loader resolved a global it should not know")
    except Exception as e:
        assert "fractions" in str(e)
    out = cd.read_mlp_checkpoint(p)
    assert out["q"] == fractions.Fraction(1, 3) and torch.equal(out["x"], torch.arange(3))


def test_legacy_file_falls_back(tmp_path):
    p = str(tmp_path / "legacy.pt")
    torch.save({"x": torch.arange(4.0)}, p, _use_new_zipfile_serialization=False)
    assert torch.equal(cd.read_mlp_checkpoint(p)["x"], torch.arange(4.0))
