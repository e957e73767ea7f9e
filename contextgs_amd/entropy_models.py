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

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as nnf

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
    def __init__(self, channel=32, init_scale=10, filters=(3, 3, 3), likelihood_bound=1e-6, tail_mass=1e-9,
                 optimize_integer_offset=True, Q=1):
        super().__init__()
        self.filters = tuple(int(t) for t in filters)
        self.init_scale = float(init_scale)
        self.likelihood_bound = float(likelihood_bound)
        self.tail_mass = float(tail_mass)
        self.optimize_integer_offset = bool(optimize_integer_offset)
        self.Q = Q
        if not 0 < self.tail_mass < 1:
            raise ValueError("`tail_mass` must be between 0 and 1")
        f = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1.0 / (len(self.filters) + 1))
        self._matrices, self._bias, self._factor = nn.ParameterList(), nn.ParameterList(), nn.ParameterList()
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1.0 / scale / f[i + 1]))
            self._matrices.append(nn.Parameter(torch.full((channel, f[i + 1], f[i]), float(init))))
            self._bias.append(nn.Parameter(torch.empty(channel, f[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self._factor.append(nn.Parameter(torch.zeros(channel, f[i + 1], 1)))

    def _logits_cumulative(self, logits, stop_gradient):
        for i in range(len(self.filters) + 1):
            matrix = nnf.softplus(self._matrices[i])
            bias = self._bias[i]
            if stop_gradient:
                matrix, bias = matrix.detach(), bias.detach()
            logits = torch.matmul(matrix, logits) + bias
            if i < len(self._factor):
                factor = torch.tanh(self._factor[i])
                if stop_gradient:
                    factor = factor.detach()
                logits = logits + factor * torch.tanh(logits)
        return logits

    def forward(self, x, Q=None):
        Q = self.Q if Q is None else Q.permute(1, 0).contiguous()
        x = x.permute(1, 0).contiguous()
        lower = self._logits_cumulative(x - 0.5 * (1 / Q), stop_gradient=False)
        upper = self._logits_cumulative(x + 0.5 * (1 / Q), stop_gradient=False)
        sign = -torch.sign(lower + upper).detach()
        likelihood = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
        bits = -torch.log2(Low_bound.apply(likelihood))
        return bits.permute(1, 0).contiguous()


class UniverseQuant(torch.autograd.Function):                # :159-171
    @staticmethod
    def forward(ctx, x):
        u = torch.empty_like(x).uniform_(-0.5, 0.5)
        return torch.round(x + u) - u

    @staticmethod
    def backward(ctx, g):
        return g
