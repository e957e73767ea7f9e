"""The C-ABI library loads (no GPU needed) and exports every symbol include/cgs.h declares;
the ctypes signature table covers the same set; host-only entry points behave."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "cgs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cgs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from contextgs_amd import _lib
    lib = _lib.lib()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/cgs.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)
    assert lib.cgs_version() >= 100


def test_no_torch_types_in_the_header():
    src = open(os.path.join(ROOT, "include", "cgs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)      # comments may cite the reference's torch call sites
    assert "torch" not in src.lower() and "at::" not in src and "#include <hip" not in src


def test_size_queries_and_errors_without_gpu():
    from contextgs_amd import _lib
    lib = _lib.lib()
    assert lib.cgs_raster_geom_bytes(1000) > 1000 * 48
    assert lib.cgs_raster_bin_bytes(1000, 5000) >= 5000 * 24
    assert lib.cgs_raster_img_bytes(1080, 1920) >= 1080 * 1920 * 8
    assert lib.cgs_sort_scratch_bytes(1 << 20) > 0 and lib.cgs_scan_scratch_bytes(1 << 20) > 0
    # argument errors are reported through the return code + cgs_last_error, never by crashing
    rc = lib.cgs_filter(None, 10, None, None, None, None, None)
    assert rc != 0 and b"cfg" in lib.cgs_last_error()
    rc = lib.cgs_ste_multistep(None, None, -1, 1, 1, None, None)
    assert rc != 0


def test_operators_refuse_cpu_tensors():
    """There is no CPU fallback behind the drop-in API."""
    import torch
    from contextgs_amd.encodings import Quantize_anchor, STE_multistep
    from contextgs_amd.entropy_models import Entropy_gaussian
    x = torch.randn(4, 3)
    with pytest.raises(RuntimeError):
        STE_multistep.apply(x, torch.ones(4, 1))
    with pytest.raises(RuntimeError):
        Quantize_anchor.apply(x, torch.zeros(1, 3), torch.ones(1, 3))
    with pytest.raises(RuntimeError):
        Entropy_gaussian()(x, x, x.abs() + 1, torch.ones(4, 1), torch.tensor(0.0))


def test_hyper_prior_likelihood_refuses_cpu_tensors():
    """EntropyBottleneck / Entropy_factorized on a CPU tensor: no silent torch fallback of the hot path."""
    import pytest
    import torch
    from contextgs_amd.entropy_bottleneck import EntropyBottleneck
    from contextgs_amd.entropy_models import Entropy_factorized
    x = torch.zeros(5, 12)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        EntropyBottleneck(12)(x, training=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Entropy_factorized(channel=12, filters=(3, 3, 3, 3))(x)
