"""Hyper-prior codec: a factorised-density `EntropyBottleneck(channels)`.

Replaces the six methods of compressai's EntropyBottleneck the reference uses
(scene/gaussian_model.py:135 ctor, :223/:913 update, :1040 quantize "symbols" +
_get_medians, :1088 compress, :1331 decompress, :1556 forward(x, training)).
compressai is NOT in the mount and unpinned (parity
unpinned); this restates the published factorised prior (Ballé et al. 2018,
the same maths as utils/entropy_models.py:103-138 `Entropy_factorized`, which
IS in the mount) and codes the symbols with our own range-ANS
(contextgs_amd.codec), so hyper.b streams are self-consistent (encode ->
decode bit-exact) but not byte-compatible with compressai's.

Layout: inputs are [N, C] (N anchors, C = feat_dim // hyper_divisor = 12
channels); every channel owns an independent 1 -> 3 -> 3 -> 3 -> 3 -> 1
cumulative-logit network.
"""
from __future__ import annotations

import math

import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F



#  update(force=True) keeps its tables while every parameter is at the (storage, version) they were built from; a
# write that bypasses the version counter (`p.data`, a collective, a raw pointer) calls invalidate_tables() — dist.broadcast_parameters
# does (load_state_dict / restore copy in place and bump the counters).
_TABLE_GENERATION = [0]


def invalidate_tables():
    _TABLE_GENERATION[0] += 1


class _LowerBound(torch.autograd.Function):
    """max(x, bound) whose gradient also passes where it moves x towards the bound's
    feasible side (the usual likelihood lower bound)."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x)
        ctx.bound = bound
        return torch.clamp(x, min=bound)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        pass_through = (x >= ctx.bound) | (g < 0)
        return g * pass_through.to(g.dtype), None


class _FusedLikelihood(torch.autograd.Function):
    """likelihood[N,C] of already-quantised values under the factorised prior (csrc/eb.hip)."""

    @staticmethod
    def forward(ctx, v, packed):
        from . import _lib
        L = _lib.lib()
        v = v.contiguous()
        packed = packed.contiguous()
        n, C = v.shape
        lik = torch.empty_like(v)
        _lib.check(L.cgs_eb_likelihood_fwd(_lib.ptr(v), _lib.ptr(packed), n, C, _lib.ptr(lik), _lib.current_stream()),
                   "cgs_eb_likelihood_fwd")
        ctx.save_for_backward(v, packed)
        return lik

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        L = _lib.lib()
        v, packed = ctx.saved_tensors
        n, C = v.shape
        g = g.contiguous()
        g_v = torch.empty_like(v)
        g_p = torch.zeros_like(packed)
        _lib.check(L.cgs_eb_likelihood_bwd(_lib.ptr(v), _lib.ptr(packed), _lib.ptr(g), n, C, _lib.ptr(g_v), _lib.ptr(g_p),
                                           _lib.current_stream()), "cgs_eb_likelihood_bwd")
        return g_v, g_p


_bits_ws = {}


class _HyperStep(torch.autograd.Function):
    """The hyper prior of one training step as ONE node: x [N,C] -> (noisy latents in coding order [N,C], summed bits
    of the rows `rows_pos` [1]) — cgs_hyper_noise_gather (x + U(-1/2,1/2) from the build's counter-based generator keyed by (seed, ORIGINAL row,
    channel), delivered in coding order) followed by cgs_eb_bits_* (scene/gaussian_model.py:1662, 1689), with the two gradient paths of the
    noisy latents (dense from the level inputs, ~15 % of the rows from the bit sum) merged here: the subset's rows are
    added into the dense gradient in place (index_add_ on distinct rows) instead of being scattered into an [N,C] zero
    buffer that autograd then adds in full."""

    @staticmethod
    def forward(ctx, x, perm, inv_perm, rows_pos, packed, seed, noisy=None, sizes=None, rows_orig=None, holder=None):
        from . import _lib
        L = _lib.lib()
        x, packed = x.contiguous(), packed.contiguous()
        _lib.require_device(x, packed)
        N, C = x.shape
        stream = _lib.current_stream()
        if noisy is not None:                 # launched earlier by noisy_latents_launch(x, perm, seed): same values
            v = noisy
            assert v.shape == x.shape and v.is_contiguous()
        else:
            v = torch.empty_like(x)
            _lib.check(L.cgs_hyper_noise_gather(_lib.ptr(x), _lib.ptr(perm), N, C, int(seed), _lib.ptr(v), stream),
                       "cgs_hyper_noise_gather")
        ws = _bits_ws.get(x.device)
        if ws is None:
            ws = _bits_ws[x.device] = torch.zeros(int(L.cgs_eb_bits_scratch_bytes()), dtype=torch.uint8, device=x.device)
        bits = torch.empty(1, dtype=torch.float32, device=x.device)
        n = int(rows_pos.shape[0]) if rows_pos is not None else N
        _lib.check(L.cgs_eb_bits_fwd(_lib.ptr(v), _lib.ptr(rows_pos), _lib.ptr(packed), n, C, _lib.ptr(ws), ws.numel(),
                                     _lib.ptr(bits), stream), "cgs_eb_bits_fwd")
        ctx.n = n
        ctx.sizes = None
        if sizes is not None and rows_pos is not None and (rows_orig is not None or perm is None) and len(sizes) <= 4:
            # sizes: the noisy latents leave as one row block per level (views of v) — their gradients then come back
            # block by block and go through the inverse permutation in ONE pass (cgs_gather_rows_segmented) instead of
            # autograd's cat of the split + a copy + a gather.  rows_orig = perm[rows_pos] (the subset's anchor rows).
            ctx.sizes = tuple(int(t) for t in sizes)
            assert sum(ctx.sizes) == N
            ctx.holder, ctx.perm = holder, perm         # ctx_ops.HyperDirect: the fused levels' backward may write g_x's rows themselves
            ctx.set_materialize_grads(False)
            ctx.save_for_backward(v, rows_pos, packed, inv_perm, rows_orig if rows_orig is not None else rows_pos)
            return (*torch.split(v, ctx.sizes), bits)
        ctx.save_for_backward(v, rows_pos, packed, inv_perm)
        return v, bits

    @staticmethod
    def _backward_blocks(ctx, *gs):
        from . import _lib
        import ctypes as C_
        L = _lib.lib()
        v, rows, packed, inv_perm, rows_orig = ctx.saved_tensors
        n, (N, C) = ctx.n, v.shape
        g_blocks, g_bits = gs[:-1], gs[-1]
        dev = v.device
        stream = _lib.current_stream()
        g_p = g_sub = None
        if g_bits is not None:
            g_sub = torch.empty(n, C, dtype=torch.float32, device=dev)
            g_p = torch.zeros_like(packed)
            _lib.check(L.cgs_eb_bits_bwd(_lib.ptr(v), _lib.ptr(rows), _lib.ptr(packed), _lib.ptr(g_bits.contiguous()), n, C,
                                         _lib.ptr(g_sub), _lib.ptr(g_p), stream), "cgs_eb_bits_bwd")
        hd = getattr(ctx, "holder", None)
        direct, done = hd.take() if hd is not None else (None, set())
        if direct is not None and not done:
            direct = None
        if all(g is None for g in g_blocks) and g_sub is None and direct is None:
            return (None,) * 10
        if direct is not None and all((sz == 0) or (j in done and g is None) for j, (g, sz) in enumerate(zip(g_blocks, ctx.sizes))):
            # every level wrote its rows of the latents' gradient itself, in parameter order (cgs_ctx_level_bwd2): nothing to gather
            g_x = direct
            if g_sub is not None:
                from .ctx_ops import add_rows_
                add_rows_(g_x, rows_orig, g_sub)
            return g_x, None, None, None, g_p, None, None, None, None, None
        srcs, lds, begin = [], [], [0]
        for g, sz in zip(g_blocks, ctx.sizes):
            if g is not None and (g.dtype != torch.float32 or g.stride(1) != 1 or (sz > 1 and g.stride(0) < C)):
                g = g.float().contiguous()
            srcs.append(g)
            lds.append(int(g.stride(0)) if (g is not None and sz > 0) else C)
            begin.append(begin[-1] + sz)
        k = len(srcs)
        g_x = torch.empty(N, C, dtype=torch.float32, device=dev)
        _lib.check(L.cgs_gather_rows_segmented(
            k, (C_.c_void_p * k)(*[None if (t is None or t.numel() == 0) else t.data_ptr() for t in srcs]),
            (C_.c_int64 * k)(*lds), (C_.c_int64 * (k + 1))(*begin), _lib.ptr(inv_perm), N, C, _lib.ptr(g_x), stream),
            "cgs_gather_rows_segmented")
        if direct is not None:          # some levels wrote their rows directly, others came back as blocks: add the direct rows
            for j in sorted(done):
                pos = torch.arange(begin[j], begin[j + 1], device=dev)
                rows_j = pos if ctx.perm is None else ctx.perm[begin[j]:begin[j + 1]]
                g_x.index_add_(0, rows_j, direct[rows_j])
        if g_sub is not None:
            from .ctx_ops import add_rows_
            add_rows_(g_x, rows_orig, g_sub)             # the rate subset's rows (distinct), in parameter order
        return g_x, None, None, None, g_p, None, None, None, None, None

    @staticmethod
    def backward(ctx, *gs):
        from . import _lib, ctx_ops
        if ctx.sizes is not None:
            return _HyperStep._backward_blocks(ctx, *gs)
        g_v, g_bits = gs
        v, rows, packed, inv_perm = ctx.saved_tensors
        n, C = ctx.n, v.shape[1]
        g_p = None
        if g_bits is not None:
            g_sub = torch.empty(n, C, dtype=torch.float32, device=v.device)
            g_p = torch.zeros_like(packed)
            _lib.check(_lib.lib().cgs_eb_bits_bwd(_lib.ptr(v), _lib.ptr(rows), _lib.ptr(packed), _lib.ptr(g_bits.contiguous()),
                                                  n, C, _lib.ptr(g_sub), _lib.ptr(g_p), _lib.current_stream()),
                       "cgs_eb_bits_bwd")
            if rows is None:
                g_v = g_sub if g_v is None else g_v + g_sub
            else:
                # never accumulate into the INCOMING gradient buffer: autograd may hand the same tensor to another
                # consumer (retain_graph, a second use of v); one [n, C] copy, the rows are added on top
                g_v = torch.zeros_like(v) if g_v is None else g_v.clone(memory_format=torch.contiguous_format)
                g_v.index_add_(0, rows, g_sub)
        if g_v is None:
            return None, None, None, None, g_p, None, None, None, None, None
        # rows of 48 bytes: torch's index_select takes its slow "vectorized gather" path for 16-byte-multiple rows
        # (210 us for [1 M, 12] on gfx950); the one-source rowcat kernel does the same gather in ~35 us
        g_x = g_v if inv_perm is None else ctx_ops.gather_rows_nograd(g_v, inv_perm)
        return g_x, None, None, None, g_p, None, None, None, None, None


class HyperBitSum:
    """What the fused training path hands the rate model instead of a likelihood tensor: the summed bits + their count."""

    def __init__(self, total, numel):
        self.total, self.numel = total, int(numel)


class _RowsOf(torch.autograd.Function):
    """x[rows] for distinct rows; backward = row scatter into zeros (no sort, unlike index_put_ accumulate)."""

    @staticmethod
    def forward(ctx, x, rows):
        ctx.save_for_backward(rows)
        ctx.shape = x.shape
        return x.index_select(0, rows)

    @staticmethod
    def backward(ctx, g):
        (rows,) = ctx.saved_tensors
        out = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        out.index_copy_(0, rows, g.contiguous())
        return out, None


def _rows_of(x, rows):
    return _RowsOf.apply(x, rows)


def cumulative_logits(matrices, biases, factors, x, stop_gradient: bool = False):
    """Cumulative logits of the factorised prior: x [C,1,M] -> [C,1,M] through per-channel layers
    logits <- softplus(M_i) @ logits + b_i (+ tanh(a_i) * tanh(logits) on all but the last).  The one torch
    statement of this density in the package: EntropyBottleneck and utils.entropy_models.Entropy_factorized both
    call it; pinned against the reference's own `_logits_cumulative` by tests/golden/entropy_api.npz."""
    take = (lambda p: p.detach()) if stop_gradient else (lambda p: p)
    logits = x
    for i, (m, b) in enumerate(zip(matrices, biases)):
        logits = torch.matmul(F.softplus(take(m)), logits) + take(b)
        if i < len(factors):
            logits = logits + torch.tanh(take(factors[i])) * torch.tanh(logits)
    return logits


def interval_likelihood(matrices, biases, factors, x, half_width, stop_gradient: bool = False):
    """P(x - h < X < x + h) = |sigmoid(s u) - sigmoid(s l)|, s = -sign(l + u) (the numerically safe tail side)."""
    lower = cumulative_logits(matrices, biases, factors, x - half_width, stop_gradient)
    upper = cumulative_logits(matrices, biases, factors, x + half_width, stop_gradient)
    sign = -torch.sign(lower + upper).detach()
    return torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))


# The density's torch statement (cumulative_logits / interval_likelihood) is HOST code by design: update() evaluates it
# on a few hundred points per channel to build the integer CDF tables of the host rANS coder (what compressai does), and
# the CPU tests pin it against the reference's own outputs.  It is NOT a fallback of the hot path: a likelihood call on a
# CPU tensor raises unless a caller (the tests) opts in explicitly.
ALLOW_HOST_FORWARD = False


def _require_device_or_host_opt_in(x, what):
    if not x.is_cuda and not ALLOW_HOST_FORWARD:
        raise RuntimeError(f"{what}: the hot path runs on device tensors through libcgs_hip.so; there is no CPU fallback "
                           "(set contextgs_amd.entropy_bottleneck.ALLOW_HOST_FORWARD to evaluate the host statement of "
                           "the density on purpose, as the golden tests do)")


_unpack_index = {}


class _PackDensity(torch.autograd.Function):
    """cat of the 14 small parameter tensors into the [C, 58] image csrc/eb.hip reads.  As plain torch ops the backward is
    14 column slices, each cloned again when it reaches its (contiguous) parameter; here it is ONE gather that lays the
    14 gradients out back to back, handed to the parameters as views."""

    @staticmethod
    def forward(ctx, channels, *parts):
        ctx.shapes = [tuple(p.shape) for p in parts]
        ctx.channels = channels
        return torch.cat([p.reshape(channels, -1) for p in parts], dim=1).contiguous()

    @staticmethod
    def backward(ctx, g):
        C = ctx.channels
        widths = [int(torch.Size(s).numel()) // C for s in ctx.shapes]
        key = (C, tuple(widths), g.device)
        idx = _unpack_index.get(key)
        if idx is None:
            total, cols, a = sum(widths), [], 0
            for w in widths:
                cols.append((torch.arange(C).unsqueeze(1) * total + a + torch.arange(w).unsqueeze(0)).reshape(-1))
                a += w
            idx = _unpack_index[key] = torch.cat(cols).to(g.device)
        flat = g.contiguous().reshape(-1).index_select(0, idx)
        out = [t.view(s) for t, s in zip(torch.split(flat, [C * w for w in widths]), ctx.shapes)]
        return (None, *out)


def pack_density_params(matrices, biases, factors, channels):
    """[C, 58] raw parameters in the order csrc/eb.hip expects (differentiable); filters (3,3,3,3) only."""
    parts = []
    for i in range(5):
        parts.append(matrices[i])
        parts.append(biases[i])
        if i < 4:
            parts.append(factors[i])
    return _PackDensity.apply(int(channels), *parts)


def fused_likelihood(v, packed):
    """likelihood [N,C] (floored at 1e-9) of values v [N,C] under the filters-(3,3,3,3) prior: one HIP launch each way."""
    return _FusedLikelihood.apply(v, packed)


class EntropyBottleneck(nn.Module):
    def __init__(self, channels: int, tail_mass: float = 1e-9, init_scale: float = 10.0,
                 filters=(3, 3, 3, 3), likelihood_bound: float = 1e-9, entropy_coder_precision: int = 16):
        super().__init__()
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        self.likelihood_bound = float(likelihood_bound)
        self.precision = int(entropy_coder_precision)

        f = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1.0 / (len(self.filters) + 1))
        self.matrices = nn.ParameterList()
        self.biases = nn.ParameterList()
        self.factors = nn.ParameterList()
        for i in range(len(self.filters) + 1):
            init = math.log(math.expm1(1.0 / scale / f[i + 1]))
            self.matrices.append(nn.Parameter(torch.full((channels, f[i + 1], f[i]), init)))
            self.biases.append(nn.Parameter(torch.empty(channels, f[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.factors.append(nn.Parameter(torch.zeros(channels, f[i + 1], 1)))
        # support: never trained in ContextGS (no aux loss anywhere) -> stays (-10, 0, 10)
        self.quantiles = nn.Parameter(torch.tensor([-self.init_scale, 0.0, self.init_scale]).repeat(channels, 1, 1))
        self.register_buffer("_offset", torch.zeros(0, dtype=torch.int32))
        self.register_buffer("_quantized_cdf", torch.zeros(0, 0, dtype=torch.int32))
        self.register_buffer("_cdf_length", torch.zeros(0, dtype=torch.int32))

    # ---- density ---------------------------------------------------------------
    def _logits_cumulative(self, x: torch.Tensor, stop_gradient: bool = False) -> torch.Tensor:
        """x: [C, 1, M] -> cumulative logits [C, 1, M]."""
        return cumulative_logits(self.matrices, self.biases, self.factors, x, stop_gradient)

    def _likelihood(self, x: torch.Tensor, stop_gradient: bool = False) -> torch.Tensor:
        return interval_likelihood(self.matrices, self.biases, self.factors, x, 0.5, stop_gradient)

    def _get_medians(self) -> torch.Tensor:
        return self.quantiles[:, :, 1:2].detach()     # [C,1,1]

    # ---- quantisation ------------------------------------------------------------
    def quantize(self, inputs: torch.Tensor, mode: str, means: torch.Tensor | None = None) -> torch.Tensor:
        if mode == "noise":
            return inputs + torch.empty_like(inputs).uniform_(-0.5, 0.5)
        out = inputs if means is None else inputs - means
        out = torch.round(out)
        if mode == "dequantize":
            return out if means is None else out + means
        if mode == "symbols":
            return out.int()
        raise ValueError(f"unknown quantisation mode {mode!r}")

    def forward(self, x: torch.Tensor, training: bool | None = None, rows: torch.Tensor | None = None):
        """x [N, C] -> (x_hat [N, C], likelihood [N, C]).

        rows (extension; distinct row indices): evaluate the likelihood of those rows only and return it as
        [len(rows), C].  The training step consumes the likelihood of the ~15 % anchors of its rate subset
        (scene/gaussian_model.py:1658-1662) and discards the rest, which is 85 % of this module's work."""
        if training is None:
            training = self.training
        assert x.dim() == 2 and x.shape[1] == self.channels, "expects [N, C]"
        if x.is_cuda and self.filters == (3, 3, 3, 3):
            # fused HIP path (csrc/eb.hip): quantisation stays a 2-op torch prologue (keeps torch's RNG stream),
            # the 1-3-3-3-3-1 density network forward/backward is one kernel each
            med = self._get_medians()[:, 0, 0]                        # [C]
            out = x + torch.empty_like(x).uniform_(-0.5, 0.5) if training else torch.round(x - med) + med
            sub = out if rows is None else out.index_select(0, rows) if not out.requires_grad else _rows_of(out, rows)
            return out, _FusedLikelihood.apply(sub, self._packed_params())
        if rows is not None:
            raise NotImplementedError("rows= is only implemented for the fused device path")
        _require_device_or_host_opt_in(x, "EntropyBottleneck.forward")
        v = x.t().reshape(self.channels, 1, -1)                      # [C,1,N]
        out = self.quantize(v, "noise" if training else "dequantize", self._get_medians())
        lik = _LowerBound.apply(self._likelihood(out), self.likelihood_bound)
        back = lambda t: t.reshape(self.channels, -1).t()
        return back(out), back(lik)

    def noisy_latents_launch(self, x: torch.Tensor, perm, seed: int):
        """Enqueue x + U(-1/2, 1/2) in coding order (the first launch of training_step_forms) ahead of time; returns
        (x, noisy, seed) to hand to training_step_forms(..., seed, noisy=noisy) of the same step."""
        from . import _lib
        assert x.is_cuda and x.dim() == 2 and x.shape[1] == self.channels and x.is_contiguous() and x.dtype == torch.float32
        v = torch.empty_like(x)
        _lib.check(_lib.lib().cgs_hyper_noise_gather(_lib.ptr(x.detach()), _lib.ptr(perm), x.shape[0], x.shape[1], int(seed),
                                                     _lib.ptr(v), _lib.current_stream()), "cgs_hyper_noise_gather")
        return x, v, int(seed)

    def training_step_forms(self, x: torch.Tensor, perm, inv_perm, rows_pos, seed: int, packed=None, noisy=None, sizes=None,
                            rows_orig=None):
        """Training-step entry (extension): (noisy latents in coding order [N,C], HyperBitSum over the coding-order
        positions rows_pos) — forward(x, training=True) restricted to what scene/gaussian_model.py:1556-1707 consumes,
        in two launches.  perm / inv_perm: the coding-order permutation and its inverse (None: identity)."""
        assert x.is_cuda and self.filters == (3, 3, 3, 3) and x.dim() == 2 and x.shape[1] == self.channels
        # packed: self._packed_params() evaluated earlier by the caller (the renderer does it before a host read-back, so
        # that the GPU has the launch queued while the host waits)
        # sizes (optional): return the noisy latents as one row block per level (a tuple) — see _HyperStep.forward
        holder = None
        if sizes is not None:
            from .ctx_ops import HyperDirect
            holder = HyperDirect(x.shape[0], x.shape[1], sizes)
        out = _HyperStep.apply(x, perm, inv_perm, rows_pos, self._packed_params() if packed is None else packed, seed, noisy,
                               sizes, rows_orig, holder)
        v_p, bits = (out[0], out[1]) if len(out) == 2 else (tuple(out[:-1]), out[-1])
        if isinstance(v_p, tuple) and holder is not None and len(v_p) == len(holder.sizes):
            for j_, blk in enumerate(v_p):               # ctx_ops.level_fused finds the holder on its block of latents
                blk._cgs_hyp_direct = (holder, j_)
        n_rows = int(rows_pos.shape[0]) if rows_pos is not None else int(x.shape[0])
        return v_p, HyperBitSum(bits, n_rows * self.channels)

    def _packed_params(self) -> torch.Tensor:
        """[C, 58] raw parameters in the order csrc/eb.hip expects (differentiable cat)."""
        return pack_density_params(self.matrices, self.biases, self.factors, self.channels)

    # ---- tables --------------------------------------------------------------------
    @torch.no_grad()
    def update(self, force: bool = False) -> bool:
        """(Re)build the per-channel quantised CDF tables used by compress/decompress."""
        if self._offset.numel() > 0 and not force:
            return False
        # (the tables are a function of the parameters alone: a forced update with every parameter at the version — and the
        #  storage — the last tables were built from rebuilds the same tables; it costs 1.4 ms and three host reads per encode)
        key = (_TABLE_GENERATION[0],) + tuple((p_.data_ptr(), p_._version) for p_ in self.parameters())
        built = getattr(self, "_tables_key", None)
        if (self._offset.numel() > 0 and built is not None and built[0] == key
                and built[1] == (self._quantized_cdf.data_ptr(), self._quantized_cdf._version, self._offset.data_ptr(), self._offset._version)):
            return True
        med = self.quantiles[:, 0, 1]
        minima = torch.clamp(torch.ceil(med - self.quantiles[:, 0, 0]), min=0).int()
        maxima = torch.clamp(torch.ceil(self.quantiles[:, 0, 2] - med), min=0).int()
        self._offset = (-minima).int()
        pmf_start = med - minima
        pmf_length = maxima + minima + 1
        max_len = int(pmf_length.max().item())
        samples = torch.arange(max_len, device=med.device, dtype=med.dtype)[None, None, :] + pmf_start[:, None, None]
        lower = self._logits_cumulative(samples - 0.5, True)
        upper = self._logits_cumulative(samples + 0.5, True)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        cdf = torch.zeros(self.channels, max_len + 2, dtype=torch.int32)
        pmf_c, tail_c, len_c = pmf.double().cpu().numpy(), tail.double().cpu().numpy(), pmf_length.cpu().numpy()
        for c in range(self.channels):
            probs = np.concatenate([pmf_c[c, : len_c[c]], tail_c[c]])
            q = pmf_to_quantized_cdf(probs, self.precision)
            cdf[c, : q.size] = torch.from_numpy(q.astype(np.int32))
        self._quantized_cdf = cdf.to(med.device)
        self._cdf_length = (pmf_length + 2).int()
        self._tables_key = (key, (self._quantized_cdf.data_ptr(), self._quantized_cdf._version, self._offset.data_ptr(), self._offset._version))
        return True

    # ---- coding (host rANS via libcgs, see codec.py) -----------------------------------
    @torch.no_grad()
    def compress(self, x: torch.Tensor) -> list[bytes]:
        """x [1, C, n] (the reference's rearrange 'b c -> 1 c b', gaussian_model.py:1088) -> [bytes]."""
        from . import codec
        assert x.dim() == 3 and x.shape[0] == 1 and x.shape[1] == self.channels
        if self._offset.numel() == 0:
            self.update()
        sym = self.quantize(x[0], "symbols", self._get_medians()[:, 0])           # [C, n] int32
        return [codec.rans_encode_channels(sym.cpu().numpy(), self._quantized_cdf.cpu().numpy(),
                                           self._cdf_length.cpu().numpy(), self._offset.cpu().numpy(),
                                           self.precision)]

    @torch.no_grad()
    def decompress(self, strings: list[bytes], size) -> torch.Tensor:
        """inverse of compress: -> [1, C, n] dequantised."""
        from . import codec
        n = int(size[0]) if isinstance(size, (tuple, list, torch.Size)) else int(size)
        if self._offset.numel() == 0:
            self.update()
        sym = codec.rans_decode_channels(strings[0], self.channels, n, self._quantized_cdf.cpu().numpy(),
                                         self._cdf_length.cpu().numpy(), self._offset.cpu().numpy(), self.precision)
        dev = self.quantiles.device
        out = torch.from_numpy(sym).to(dev).to(self.quantiles.dtype) + self._get_medians()[:, 0]
        return out.unsqueeze(0)


    # The container codes the hyper latents as independent 10 000-anchor rANS strings
    # (scene/gaussian_model.py:1082-1098, 1326-1336).  The strings are serial inside and independent of each other,
    # so the chunk forms below quantise / dequantise ONCE on the device and run the strings concurrently on host
    # threads (the ctypes calls into libcgs release the GIL).  Byte-identical to a loop of compress()/decompress().
    @torch.no_grad()
    def compress_chunks(self, x: torch.Tensor, chunk: int, lazy: bool = False) -> list[bytes]:
        """x [C, N] -> [compress(x[None, :, s:s+chunk])[0] for s in range(0, N, chunk)].
        lazy: return the host-thread futures instead of waiting for them (the caller keeps the device busy meanwhile)."""
        from . import codec
        assert x.dim() == 2 and x.shape[0] == self.channels
        if self._offset.numel() == 0:
            self.update()
        sym_d = self.quantize(x, "symbols", self._get_medians()[:, 0])                        # [C, N] int32
        sym = codec.to_host_pinned(sym_d, "hyper symbols") if (lazy and sym_d.is_cuda) else sym_d.cpu().numpy()
        tabs = (self._quantized_cdf.cpu().numpy(), self._cdf_length.cpu().numpy(), self._offset.cpu().numpy())
        jobs = [codec.host_pool().submit(codec.rans_encode_channels, sym[:, s:s + chunk], *tabs, self.precision)
                for s in range(0, sym.shape[1], chunk)]
        return jobs if lazy else [j.result() for j in jobs]

    @torch.no_grad()
    def decompress_chunks_rows(self, strings: list[bytes], sizes: list[int], tasks: int = 0):
        """decompress_chunks(...).t() as a job: the chunk strings are decoded by `tasks` host-thread tasks (a contiguous
        run of chunks each: few, long tasks keep the pool off the GIL the launching thread needs) straight into the
        rows of one pinned [N, C] float buffer, already dequantised (symbol + median, the same fp32 sum).  Returns a
        callable that waits for the tasks and returns the [N, C] tensor on the module's device (one H2D copy)."""
        from . import _lib, codec
        import ctypes as C_
        if self._offset.numel() == 0:
            self.update()
        dev = self.quantiles.device
        Cn, N = self.channels, int(sum(sizes))
        if not strings:
            return lambda: torch.zeros(0, Cn, dtype=self.quantiles.dtype, device=dev)
        cdf = np.ascontiguousarray(self._quantized_cdf.cpu().numpy(), dtype=np.int32)
        cl = np.ascontiguousarray(self._cdf_length.cpu().numpy(), dtype=np.int32)
        of = np.ascontiguousarray(self._offset.cpu().numpy(), dtype=np.int32)
        med = np.ascontiguousarray(self._get_medians()[:, 0, 0].float().cpu().numpy(), dtype=np.float32)
        stage = _rows_staging(N * Cn)
        base = stage.data_ptr()
        L = _lib.lib()
        starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        ptr = lambda a: a.ctypes.data_as(C_.c_void_p)

        def run(lo, hi):
            for k in range(lo, hi):
                buf = np.frombuffer(strings[k], dtype=np.uint8)
                _lib.check(L.cgs_rans_decode_rows_host(ptr(buf), buf.size, Cn, int(sizes[k]), ptr(cdf), cdf.shape[1], ptr(cl),
                                                       ptr(of), self.precision, ptr(med), base + 4 * Cn * int(starts[k]), Cn),
                           "cgs_rans_decode_rows_host")

        n_chunks = len(strings)
        tasks = tasks or int(os.environ.get("CGS_HYPER_TASKS", "32"))
        per = -(-n_chunks // max(1, min(tasks, n_chunks)))
        jobs = [codec.host_pool().submit(run, lo, min(n_chunks, lo + per)) for lo in range(0, n_chunks, per)]

        def finish():
            for j in jobs:
                j.result()
            return stage[:N * Cn].view(N, Cn).to(dev, non_blocking=True).to(self.quantiles.dtype)
        return finish

    # Container version 2: the same symbols and tables through the DEVICE coder (csrc/codec.hip, "Lane-parallel table
    # codec"): blocks of `block` anchors of one channel, 64 interleaved lane streams each — no host rANS strings.
    @torch.no_grad()
    def compress_lanes(self, x: torch.Tensor, block: int):
        """x [C, N] (device) -> (blob uint8 ndarray, lens int64 [blocks])."""
        from . import codec
        assert x.dim() == 2 and x.shape[0] == self.channels
        if self._offset.numel() == 0:
            self.update()
        sym = self.quantize(x, "symbols", self._get_medians()[:, 0])                          # [C, N] int32
        return codec.table_encode_lanes(sym, self._quantized_cdf, self._cdf_length, self._offset, block)

    @torch.no_grad()
    def decompress_lanes_rows(self, blob, lens, n: int, block: int) -> torch.Tensor:
        """inverse of compress_lanes -> [n, C] dequantised rows on the module's device."""
        from . import codec
        if self._offset.numel() == 0:
            self.update()
        med = self._get_medians()[:, 0, 0].float()
        return codec.table_decode_lanes(blob, lens, self.channels, int(n), self._quantized_cdf, self._cdf_length, self._offset,
                                        med, block).to(self.quantiles.dtype)

    @torch.no_grad()
    def decompress_chunks(self, strings: list[bytes], sizes: list[int]) -> torch.Tensor:
        """inverse of compress_chunks -> [C, sum(sizes)] dequantised, on the module's device."""
        from . import codec
        if self._offset.numel() == 0:
            self.update()
        dev = self.quantiles.device
        if not strings:
            return torch.zeros(self.channels, 0, dtype=self.quantiles.dtype, device=dev)
        tabs = (self._quantized_cdf.cpu().numpy(), self._cdf_length.cpu().numpy(), self._offset.cpu().numpy())
        jobs = [codec.host_pool().submit(codec.rans_decode_channels, b, self.channels, int(n), *tabs, self.precision)
                for b, n in zip(strings, sizes)]
        sym = np.concatenate([j.result() for j in jobs], axis=1)
        return torch.from_numpy(sym).to(dev).to(self.quantiles.dtype) + self._get_medians()[:, 0]


_ROWS_STAGE = None


def _rows_staging(n_floats: int) -> torch.Tensor:
    """Grow-only pinned float buffer the hyper chunk jobs decode into (one H2D copy afterwards)."""
    global _ROWS_STAGE
    if _ROWS_STAGE is None or _ROWS_STAGE.numel() < n_floats:
        _ROWS_STAGE = torch.empty(int(n_floats * 1.25) + 1024, dtype=torch.float32, pin_memory=torch.cuda.is_available())
    return _ROWS_STAGE


def pmf_to_quantized_cdf(pmf: np.ndarray, precision: int = 16) -> np.ndarray:
    """Probabilities -> strictly increasing integer CDF with total 2**precision
    (every symbol keeps frequency >= 1; the excess is taken from the symbols that lose
    the least relative mass).  Deterministic pure-integer steal loop."""
    total = 1 << precision
    freq = np.maximum(1, np.round(np.asarray(pmf, dtype=np.float64) / max(pmf.sum(), 1e-300) * total)).astype(np.int64)
    diff = int(freq.sum() - total)
    while diff != 0:
        if diff > 0:      # remove from the largest entries first
            i = int(np.argmax(freq))
            take = min(diff, int(freq[i]) - 1)
            if take <= 0:
                raise ValueError("cannot normalise pmf")
            freq[i] -= take
            diff -= take
        else:
            i = int(np.argmax(freq))
            freq[i] += -diff
            diff = 0
    cdf = np.zeros(freq.size + 1, dtype=np.int64)
    np.cumsum(freq, out=cdf[1:])
    assert cdf[-1] == total and (np.diff(cdf) > 0).all()
    return cdf
