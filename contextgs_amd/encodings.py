"""Drop-in for the reference's `utils/encodings.py`: quantisers (STE_*,
Quantize_anchor), the Gaussian / Bernoulli codecs and the size estimate, with
the same names, signatures and return values.  Arithmetic runs in
libcgs_hip.so (elementwise.hip, codec kernels) and its host coder; nothing
here falls back to a CPU implementation.

Cited lines are utils/encodings.py.
"""
from __future__ import annotations

import torch

from . import _lib

anchor_round_digits = 16                     # :10
Q_anchor = 1 / (2 ** anchor_round_digits - 1)  # :11
use_clamp = True                             # :12
use_multiprocessor = False                   # :13  (the multiprocess variants are dead code in the reference)


def _c(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def _q_layout(x: torch.Tensor, Q: torch.Tensor):
    """Return (Q_flat, q_div) such that element i of x.flatten() uses Q_flat[i // q_div].
    Supports the layouts the reference uses: same shape, [n,1] vs [n,c], [n,1,1] vs [n,k,3],
    or a single element."""
    if Q.numel() == 1:
        return _c(Q).reshape(1), max(1, x.numel())
    if Q.shape == x.shape:
        return _c(Q).reshape(-1), 1
    if Q.dim() == x.dim() and Q.shape[0] == x.shape[0] and Q.numel() == x.shape[0]:
        return _c(Q).reshape(-1), max(1, x.numel() // max(1, x.shape[0]))
    return _c(Q.expand_as(x)).reshape(-1), 1


class STE_binary(torch.autograd.Function):                  # :183-200
    @staticmethod
    def forward(ctx, input):
        ctx.save_for_backward(input)
        return torch.where(torch.clamp(input, -1, 1) >= 0, 1.0, -1.0).to(input.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        (inp,) = ctx.saved_tensors
        return grad_output * ((inp >= -1) & (inp <= 1)).to(grad_output.dtype)


class STE_multistep(torch.autograd.Function):               # :203-216
    @staticmethod
    def forward(ctx, input, Q):
        _lib.require_device(input)
        if not isinstance(Q, torch.Tensor):
            Q = torch.tensor([float(Q)], dtype=torch.float32, device=input.device)
        x = _c(input)
        qf, q_div = _q_layout(x, Q.to(x.device))
        out = torch.empty_like(x)
        _lib.check(_lib.lib().cgs_ste_multistep(_lib.ptr(x), _lib.ptr(qf), x.numel(), q_div, int(use_clamp),
                                                _lib.ptr(out), _lib.current_stream()), "cgs_ste_multistep")
        return out

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output, None


class Quantize_anchor(torch.autograd.Function):             # :219-231
    @staticmethod
    def forward(ctx, anchors, min_v, max_v):
        _lib.require_device(anchors, min_v, max_v)
        a = _c(anchors)
        lo, hi = _c(min_v).reshape(-1), _c(max_v).reshape(-1)
        assert a.dim() == 2 and a.shape[1] == 3 and lo.numel() == 3 and hi.numel() == 3
        aq, qv = torch.empty_like(a), torch.empty_like(a)
        _lib.check(_lib.lib().cgs_quantize_anchor(_lib.ptr(a), _lib.ptr(lo), _lib.ptr(hi), a.shape[0],
                                                  anchor_round_digits, _lib.ptr(aq), _lib.ptr(qv),
                                                  _lib.current_stream()), "cgs_quantize_anchor")
        ctx.mark_non_differentiable(qv)
        ctx.set_materialize_grads(False)       # no zeros [N,3] for the gradient of the non-differentiable second output
        return aq, qv

    @staticmethod
    def backward(ctx, grad_output, _tmp):
        return grad_output, None, None


class Quantize_anchor_attach(torch.autograd.Function):
    """Quantize_anchor's node around values that an earlier call of the same step computed from the same parameter (model.get_anchor:
    the anchor cull of a step quantises the anchors without a graph, render() quantises them again with one): no launch, the same
    straight-through backward."""
    @staticmethod
    def forward(ctx, anchors, values):
        return values.view_as(values)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output, None


def get_binary_vxl_size(binary_vxl):                         # :15-32
    ttl_num = binary_vxl.numel()
    pos_num = torch.sum(binary_vxl)
    neg_num = ttl_num - pos_num
    Pg = torch.clamp(pos_num / ttl_num, min=1e-6, max=1 - 1e-6)
    ttl_bit = pos_num * (-torch.log2(Pg)) + neg_num * (-torch.log2(1 - Pg))
    ttl_bit = ttl_bit + 32
    return Pg, ttl_bit, ttl_bit.item() / 8.0 / 1024 / 1024, ttl_num


# ---- codecs (implemented in codec.py; re-exported under the reference's names) ----
def encoder_gaussian(x, mean, scale, Q, file_name=None):      # :83-116
    from .codec import encoder_gaussian as f
    return f(x, mean, scale, Q, file_name)


def decoder_gaussian(mean, scale, Q, file_name=None, min_value=-100, max_value=100, bstream=None):   # :119-144
    from .codec import decoder_gaussian as f
    return f(mean, scale, Q, file_name, min_value, max_value, bstream)


def encoder(x, p, file_name):                                  # :147-163
    from .codec import encoder as f
    return f(x, p, file_name)


def decoder(p, file_name):                                     # :165-180
    from .codec import decoder as f
    return f(p, file_name)
