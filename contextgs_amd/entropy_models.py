"""Drop-in for the reference's `utils/entropy_models.py`: same six classes,
same constructor / forward signatures.  `Entropy_gaussian` (the live one) and
`Entropy_gaussian_clamp` run as one fused HIP kernel forward and one backward
(elementwise.hip), which also removes the reference's GPU->CPU->GPU numpy
round trip in Low_bound.backward (:153-155).  The dead-but-public classes
(`Entropy_bernoulli`, `Entropy_factorized`, `UniverseQuant`) are thin torch
compositions kept for API parity.

Cited lines are utils/entropy_models.py.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from . import encodings as _enc
from .encodings import _c, _q_layout


class _GaussianRate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mean, scale, Q, x_mean, clamp):
        _lib.require_device(x, mean, scale)
        shape = x.shape
        xc, mc, sc = _c(x), _c(mean.expand(shape)), _c(scale.expand(shape))
        qf, q_div = _q_layout(xc, Q)
        xm = _c(x_mean.detach()).reshape(1) if clamp else None
        bits = torch.empty_like(xc)
        _lib.check(_lib.lib().cgs_entropy_gaussian_fwd(_lib.ptr(xc), _lib.ptr(mc), _lib.ptr(sc), _lib.ptr(qf),
                                                       xc.numel(), q_div, _lib.ptr(xm), int(clamp), _lib.ptr(bits),
                                                       _lib.current_stream()), "cgs_entropy_gaussian_fwd")
        ctx.save_for_backward(xc, mc, sc, qf, xm if clamp else torch.empty(0, device=xc.device))
        ctx.q_div, ctx.clamp = q_div, clamp
        ctx.shapes = (x.shape, mean.shape, scale.shape, Q.shape)
        return bits

    @staticmethod
    def backward(ctx, g):
        xc, mc, sc, qf, xm = ctx.saved_tensors
        g = _c(g)
        gx, gm, gs, gq = (torch.empty_like(xc) for _ in range(4))
        _lib.check(_lib.lib().cgs_entropy_gaussian_bwd(
            _lib.ptr(xc), _lib.ptr(mc), _lib.ptr(sc), _lib.ptr(qf), xc.numel(), ctx.q_div,
            _lib.ptr(xm) if ctx.clamp else None, int(ctx.clamp), _lib.ptr(g), _lib.ptr(gx), _lib.ptr(gm),
            _lib.ptr(gs), _lib.ptr(gq), _lib.current_stream()), "cgs_entropy_gaussian_bwd")
        xs, ms, ss, qs = ctx.shapes
        return (gx.sum_to_size(xs), gm.sum_to_size(ms), gs.sum_to_size(ss), gq.sum_to_size(qs) if len(qs) else gq.sum(),
                None, None)


def _rate(x, mean, scale, Q, x_mean):
    if not isinstance(Q, torch.Tensor):
        Q = torch.tensor(float(Q), dtype=torch.float32, device=x.device)
    if _enc.use_clamp and x_mean is None:
        x_mean = x.mean()
    if _enc.use_clamp and not isinstance(x_mean, torch.Tensor):
        x_mean = torch.tensor(float(x_mean), dtype=torch.float32, device=x.device)
    return _GaussianRate.apply(x, mean, scale, Q, x_mean, bool(_enc.use_clamp))


class Entropy_gaussian_clamp(nn.Module):                     # :8-27
    def __init__(self, Q=1):
        super().__init__()
        self.Q = Q

    def forward(self, x, mean, scale, Q=None):
        return _rate(x, mean, scale, self.Q if Q is None else Q, None)


class Entropy_gaussian(nn.Module):                           # :30-50
    def __init__(self, Q=1):
        super().__init__()
        self.Q = Q

    def forward(self, x, mean, scale, Q=None, x_mean=None):
        return _rate(x, mean, scale, self.Q if Q is None else Q, x_mean)


class Entropy_bernoulli(nn.Module):                          # :53-64
    def forward(self, x, p):
        p = torch.clamp(p, min=1e-6, max=1 - 1e-6)
        pos_mask = (1 + x) / 2.0
        neg_mask = (1 - x) / 2.0
        return -torch.log2(p) * pos_mask + -torch.log2(1 - p) * neg_mask


class Low_bound(torch.autograd.Function):                    # :141-156 (device-side; no host round trip)
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.clamp(x, min=1e-6)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        grad1 = g.clone()
        grad1[x < 1e-6] = 0
        return grad1 * ((x >= 1e-6) | (g < 0.0)).to(g.dtype)


class Entropy_factorized(nn.Module):                         # :67-138
    """Fully factorised learned prior, one density network per channel (never instantiated by ContextGS, kept for
    API parity).  Same constructor, parameter lists (`_matrices`, `_bias`, `_factor`) and initialisation as the
    reference; the density itself is NOT restated here: `_logits_cumulative` and `forward` call the package's one
    implementation (entropy_bottleneck.cumulative_logits / interval_likelihood; on the device with filters
    (3,3,3,3) and Q = 1 the fused kernel of csrc/eb.hip).

    forward(x [N,C], Q) -> bits [N,C] = -log2(max(P(x - 1/(2Q) < X < x + 1/(2Q)), likelihood_bound)).  The
    reference's forward (:121-138) cannot be pinned: it raises for every input (its `[C,N]` tensor meets `[C,f,1]`
    matrices in a matmul without the singleton axis; with channel=1 the final `permute(1, 0)` is applied to a
    3-D tensor) — this is the [N,C] semantics its comments state.  The part that does run there,
    `_logits_cumulative` on `[C,1,M]`, is what tests/golden/entropy_api.npz pins."""

    def __init__(self, channel=32, init_scale=10, filters=(3, 3, 3), likelihood_bound=1e-6, tail_mass=1e-9,
                 optimize_integer_offset=True, Q=1):
        super().__init__()
        from .entropy_bottleneck import EntropyBottleneck
        self.filters = tuple(int(t) for t in filters)
        self.init_scale, self.likelihood_bound, self.tail_mass = float(init_scale), float(likelihood_bound), float(tail_mass)
        self.optimize_integer_offset = bool(optimize_integer_offset)
        self.Q = Q
        self.channel = int(channel)
        if not 0 < self.tail_mass < 1:
            raise ValueError("`tail_mass` must be between 0 and 1")
        # the bottleneck's constructor already builds this exact parameter set (same shapes, same init rule):
        # adopt its lists under the reference's attribute names
        proto = EntropyBottleneck(self.channel, tail_mass=self.tail_mass, init_scale=self.init_scale, filters=self.filters)
        self._matrices, self._bias, self._factor = proto.matrices, proto.biases, proto.factors

    def _logits_cumulative(self, logits, stop_gradient):
        from .entropy_bottleneck import cumulative_logits
        return cumulative_logits(self._matrices, self._bias, self._factor, logits, stop_gradient)

    def forward(self, x, Q=None):
        from . import entropy_bottleneck as eb
        assert x.dim() == 2 and x.shape[1] == self.channel, "expects [N, C]"
        Q = self.Q if Q is None else Q
        unit_q = not isinstance(Q, torch.Tensor) and float(Q) == 1.0
        if x.is_cuda and self.filters == (3, 3, 3, 3) and unit_q:
            lik = eb.fused_likelihood(_c(x), eb.pack_density_params(self._matrices, self._bias, self._factor, self.channel))
        else:
            eb._require_device_or_host_opt_in(x, "Entropy_factorized.forward")
            half = 0.5 / Q if not isinstance(Q, torch.Tensor) else (0.5 / Q.expand(x.shape)).t().reshape(self.channel, 1, -1)
            v = x.t().reshape(self.channel, 1, -1)
            lik = eb.interval_likelihood(self._matrices, self._bias, self._factor, v, half).reshape(self.channel, -1).t()
        return -torch.log2(Low_bound.apply(lik) if self.likelihood_bound == 1e-6
                           else eb._LowerBound.apply(lik, self.likelihood_bound))


class UniverseQuant(torch.autograd.Function):                # :159-171
    """round(x + u) - u with u ~ U(-1/2, 1/2) drawn per element (subtractive dither); identity gradient."""

    @staticmethod
    def forward(ctx, x):
        u = torch.rand_like(x) - 0.5
        return torch.round(x + u) - u

    @staticmethod
    def backward(ctx, g):
        return g
