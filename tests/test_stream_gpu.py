"""_lib.current_stream() is torch's current stream — also inside a torch.cuda.stream(...) block (the library enqueues there)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_current_stream_follows_torch():
    from contextgs_amd import _lib
    assert _lib.current_stream() == torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert _lib.current_stream() == side.cuda_stream == torch.cuda.current_stream().cuda_stream
        # and work issued through the library lands on it: a scan on the side stream, ordered by that stream alone
        x = torch.ones(1 << 16, dtype=torch.int32, device="cuda")
        out = torch.empty_like(x)
        ws = torch.empty(int(_lib.lib().cgs_scan_scratch_bytes(x.numel())), dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().cgs_scan_exclusive_u32(_lib.ptr(x), _lib.ptr(out), x.numel(), _lib.ptr(ws), ws.numel(),
                                                      _lib.current_stream()), "cgs_scan_exclusive_u32")
        side.synchronize()
        assert int(out[-1]) == x.numel() - 1
    assert _lib.current_stream() == torch.cuda.current_stream().cuda_stream
