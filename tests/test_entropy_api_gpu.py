"""GPU parity of the hyper prior's density kernel (csrc/eb.hip) and of the dead-but-public
`utils.entropy_models` / `utils.encodings` API (b5, b8) against tests/golden/entropy_api.npz — outputs of the
REFERENCE's own `Entropy_factorized._logits_cumulative` + sigmoid-difference likelihood
(utils/entropy_models.py:103-135), `Entropy_gaussian_clamp`, `UniverseQuant` and `get_binary_vxl_size`."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _bottleneck(seed):
    from contextgs_amd.entropy_bottleneck import EntropyBottleneck
    eb = EntropyBottleneck(gi.H).cuda()
    W = gi.mlp_weights(seed)
    with torch.no_grad():
        for i in range(5):
            eb.matrices[i].copy_(T(W[f"latent_codec.matrices.{i}"]))
            eb.biases[i].copy_(T(W[f"latent_codec.biases.{i}"]))
            if i < 4:
                eb.factors[i].copy_(T(W[f"latent_codec.factors.{i}"]))
    return eb


@pytest.mark.parametrize("seed", [2, 7])
def test_density_kernel_matches_reference_density(seed):
    """cgs_eb_likelihood_{fwd,bwd} vs the reference's density network: likelihood values, d/dvalue and the
    gradient of all 58 parameters per channel (the reference's autograd through its own `_logits_cumulative`)."""
    from contextgs_amd.entropy_bottleneck import fused_likelihood
    g = np.load(os.path.join(GOLD, "entropy_api.npz"))
    eb = _bottleneck(seed)
    v = T(gi.factorized_inputs(seed)).requires_grad_(True)
    lik = fused_likelihood(v, eb._packed_params())
    ref = g[f"fz{seed}_lik"]
    assert ref.min() > 1e-6                                   # no bounded entry: bounds are not part of this comparison
    assert np.abs(lik.detach().cpu().numpy() - ref).max() <= 3e-7        # difference of two fp32 sigmoids
    (lik * T(g[f"fz{seed}_gw"])).sum().backward()
    gv = g[f"fz{seed}_gv"]
    assert np.abs(v.grad.cpu().numpy() - gv).max() <= 2e-4 * np.abs(gv).max()
    for i in range(5):
        for got, key in ((eb.matrices[i].grad, f"fz{seed}_g_matrices.{i}"), (eb.biases[i].grad, f"fz{seed}_g_biases.{i}")) + \
                (((eb.factors[i].grad, f"fz{seed}_g_factors.{i}"),) if i < 4 else ()):
            r = g[key]
            err = np.abs(got.cpu().numpy() - r).max()
            assert err <= 3e-4 * max(1e-6, np.abs(r).max()), (key, err, np.abs(r).max())


@pytest.mark.parametrize("seed", [2, 7])
def test_bottleneck_forward_eval_and_factorized_class(seed):
    from contextgs_amd.entropy_models import Entropy_factorized
    g = np.load(os.path.join(GOLD, "entropy_api.npz"))
    eb = _bottleneck(seed)
    v = gi.factorized_inputs(seed)
    # the values are not integers: eval quantisation rounds them; the likelihood of round(v) must equal the density
    # kernel evaluated at round(v) -> compare on the rows that already hold integers (rows 0..21 of channel 0)
    out, lik = eb(T(v), training=False)
    assert torch.equal(out, torch.round(T(v)))
    ints = np.arange(-10, 11, dtype=np.float32)
    assert np.abs(lik[:21, 0].detach().cpu().numpy() - g[f"fz{seed}_lik"][:21, 0]).max() <= 3e-7
    m = Entropy_factorized(channel=gi.H, filters=(3, 3, 3, 3)).cuda()
    with torch.no_grad():
        for i in range(5):
            m._matrices[i].copy_(eb.matrices[i])
            m._bias[i].copy_(eb.biases[i])
            if i < 4:
                m._factor[i].copy_(eb.factors[i])
    bits = m(T(v))                                            # fused kernel, bound 1e-6
    assert np.abs(bits.detach().cpu().numpy() - g[f"fz{seed}_bits"]).max() <= 1e-4
    bits_q = m(T(v), torch.ones(v.shape[0], gi.H, device="cuda"))      # tensor Q = 1 -> torch composition, same numbers
    assert np.abs(bits_q.detach().cpu().numpy() - g[f"fz{seed}_bits"]).max() <= 1e-4
    assert m._logits_cumulative(T(v).t().reshape(gi.H, 1, -1) - 0.5, False).shape == (gi.H, 1, v.shape[0])


def test_entropy_gaussian_clamp_matches_reference():
    from contextgs_amd.entropy_models import Entropy_gaussian_clamp
    g = np.load(os.path.join(GOLD, "entropy_api.npz"))
    x, mean, scale, Q = gi.elementwise_inputs(193, 6)
    xg, mg, sg, Qg = (T(v).requires_grad_(True) for v in (x, mean, scale, Q))
    bits = Entropy_gaussian_clamp(Q=1)(xg, mg, sg, Qg)
    (bits * T(g["egc_gw"])).sum().backward()
    b = bits.detach().cpu().numpy()
    assert np.abs(np.exp2(-b) - np.exp2(-g["egc_bits"])).max() <= 3e-7 and np.abs(b - g["egc_bits"]).max() <= 0.1
    well = g["egc_bits"] < 10
    for a, ref in ((xg.grad, g["egc_gx"]), (mg.grad, g["egc_gmean"]), (sg.grad, g["egc_gscale"])):
        a = a.cpu().numpy()
        assert np.allclose(a[well], ref[well], rtol=2e-3, atol=1e-5 * np.abs(ref).max())
        assert np.allclose(a[~well], ref[~well], rtol=0.15, atol=1e-3 * np.abs(ref).max())
    assert np.allclose(Qg.grad.cpu().numpy(), g["egc_gQ"], rtol=0.05, atol=1e-2 * np.abs(g["egc_gQ"]).max())
    b2 = Entropy_gaussian_clamp(Q=0.25)(T(x), T(mean), T(scale)).cpu().numpy()
    assert np.abs(np.exp2(-b2) - np.exp2(-g["egc_bits_scalarQ"])).max() <= 3e-7


def test_universe_quant_statistics_and_vxl_size():
    from contextgs_amd.encodings import get_binary_vxl_size
    from contextgs_amd.entropy_models import UniverseQuant
    g = np.load(os.path.join(GOLD, "entropy_api.npz"))
    torch.manual_seed(5)
    x = T(np.random.default_rng(20).normal(0, 3, size=(400, 250)).astype(np.float32)).requires_grad_(True)
    y = UniverseQuant.apply(x)
    y.sum().backward()
    e = (y - x).detach().double()
    # 1e5 samples of a U(-1/2, 1/2) error: mean within 4 sigma (sigma = 0.2887 / sqrt(1e5) = 9e-4), variance 1/12
    assert abs(float(e.mean())) < 4e-3 and abs(float(e.var()) - float(g["uq_err_var"])) < 1.5e-3
    assert float(e.abs().max()) <= 0.5 + 1e-6 and bool((x.grad == 1).all())
    # round(x + u) is an integer
    rng = np.random.default_rng(21)
    for k, (n, p1) in enumerate(((1000, 0.7), (30, 0.0), (30, 1.0), (77777, 0.013))):
        m = (rng.random((n, 10, 1)) < p1).astype(np.float32)
        Pg, ttl_bit, mb, ttl_num = get_binary_vxl_size(T(m))
        ref = g[f"bvs_{k}"]
        assert np.allclose([Pg.item(), ttl_bit.item(), mb, ttl_num], ref[:4], rtol=2e-6), (k, ref)


def test_mask_stream_coder_from_device_tensors(tmp_path):
    """b8: `encoder` / `decoder` (utils/encodings.py:147-180) called the way conduct_encoding does, with DEVICE
    tensors.  The single Bernoulli stream is coded by the library's host coder (cgs_ac_*_const_host: one serial
    stream has no device parallelism); its bytes must equal the oracle coder's on the same 16-bit table."""
    from contextgs_amd.encodings import decoder, encoder
    from oracle import codec_ref as ref
    rng = np.random.default_rng(8)
    for n, p1 in ((30000, 0.7), (1, 0.5), (4097, 0.02)):
        x = T(np.where(rng.random(n) < p1, 1.0, -1.0).astype(np.float32))
        p = torch.full_like(x, float((x > 0).float().mean()))
        f = str(tmp_path / f"masks_{n}.b")
        bits = encoder(x, p, f)
        assert bits == 8 * os.path.getsize(f)
        back = decoder(p, f)
        assert back.is_cuda and torch.equal(back, x)
        row = ref.float_cdf_to_int([0.0, 1 - float(p[0]), 1.0])
        assert open(f, "rb").read() == ref.ac_encode([row] * n, ((x + 1) / 2).to(torch.int64).tolist())
