"""The rasterizer's workspace pool (contextgs_amd/rasterizer.py: _workspace): a pooled buffer goes back to its size class only
when the tensor handed out is really dead — not while an autograd node still holds it through save_for_backward after the Python
name went out of scope — and the next request of that class gets the same memory (no growth)."""
import gc

import torch


def test_pooled_workspace_outlives_its_python_name_inside_autograd_and_is_reused():
    from contextgs_amd import rasterizer as rz
    rz.workspace_pool_clear()

    class Keep(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            ws = rz._workspace(3_000_000, "cpu")
            ws[:8] = 7
            ctx.save_for_backward(ws)
            ctx.ptr = ws.data_ptr()
            return x * 2

        @staticmethod
        def backward(ctx, g):
            (ws,) = ctx.saved_tensors
            assert ws.data_ptr() == ctx.ptr and int(ws[0]) == 7 and ws.numel() == 3_000_000
            return g * 2

    x = torch.ones(4, requires_grad=True)
    y = Keep.apply(x)
    gc.collect()
    key = ("cpu", rz._size_class(3_000_000))
    assert not rz._POOL_FREE.get(key), "the buffer was returned while the autograd node still holds it"
    other = rz._workspace(3_000_000, "cpu")          # a second live request of the class: another buffer
    other[:8] = 1
    y.sum().backward()
    ptr = y.grad_fn.ptr if y.grad_fn is not None else None
    del y
    gc.collect()
    assert len(rz._POOL_FREE.get(key, [])) == 1
    again = rz._workspace(2_900_000, "cpu")          # same 1/8-octave class: the pooled buffer, not a new one
    assert rz._size_class(2_900_000) == rz._size_class(3_000_000)
    assert again.data_ptr() == ptr and again.numel() == 2_900_000
    del other, again
    gc.collect()
    assert len(rz._POOL_FREE[key]) == 2
    rz.workspace_pool_clear()


def test_size_classes_are_eighth_octaves():
    from contextgs_amd.rasterizer import _size_class
    assert _size_class(1) == 1 << 20 and _size_class((1 << 20) + 1) == (1 << 20) + (1 << 17)
    for n in (5_000_001, 580_000_000, 4_150_000_000):
        c = _size_class(n)
        assert n <= c <= n * 1.14 and _size_class(c) == c
