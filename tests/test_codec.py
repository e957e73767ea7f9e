"""Entropy coders on the host (no GPU needed): product (libcgs_hip.so's C-ABI host entry
points, via contextgs_amd.codec) vs the pure-Python oracle (oracle/codec_ref.py), plus
hand-computed known-answer streams.  Bit-exact everywhere."""
import numpy as np
import pytest
import torch

from contextgs_amd import codec
from oracle import codec_ref as ref


def _rows(rng, n, Lp):
    pm = rng.random((n, Lp - 1)).astype(np.float32) ** 3 + 1e-4
    pm /= pm.sum(1, keepdims=True)
    cdf = np.concatenate([np.zeros((n, 1), np.float32), np.cumsum(pm, 1, dtype=np.float32)], 1)
    return np.clip(cdf, 0, 1).astype(np.float32)


def test_known_answer_single_symbols():
    # Lp = 3, symbols {0,1}, P(0) = 1/2: hand computation (see oracle/codec_ref.py docstring maths):
    # symbol 0 -> bits 0,0,1 -> 0x20 ; symbol 1 -> bits 1,0,1 -> 0xA0
    row = [[0, 32768, 0]]
    assert ref.ac_encode(row, [0]) == b"\x20"
    assert ref.ac_encode(row, [1]) == b"\xa0"
    cdf = torch.tensor([[0.0, 32768 / 65534 - 1 / 65534, 1.0]])     # -> int row [0, 32768, 0]
    assert ref.float_cdf_to_int(cdf[0].numpy()) == [0, 32768, 0]
    assert codec.encode_float_cdf(cdf, torch.tensor([0], dtype=torch.int16)) == b"\x20"
    assert codec.encode_float_cdf(cdf, torch.tensor([1], dtype=torch.int16)) == b"\xa0"
    # two symbols "0,1": interval [0.25, 0.5) -> bits 0,1 then terminator: low=0 after renorm -> 0 + pending 1
    assert ref.ac_encode(row * 2, [0, 1]) == codec.encode_float_cdf(cdf.repeat(2, 1), torch.tensor([0, 1], dtype=torch.int16))
    assert ref.ac_encode(row * 2, [0, 1]) == bytes([0b01010000])


@pytest.mark.parametrize("n,Lp,seed", [(1, 2, 0), (17, 3, 1), (400, 9, 2), (300, 64, 3), (50, 700, 4)])
def test_table_coder_matches_oracle_bit_exact(n, Lp, seed):
    rng = np.random.default_rng(seed)
    cdf = _rows(rng, n, Lp)
    sym = rng.integers(0, Lp - 1, size=n).astype(np.int16)
    sym[0] = Lp - 2                       # the top symbol (upper bound 2^16) is always exercised
    rows = [ref.float_cdf_to_int(r) for r in cdf]
    got_rows = codec._cdf_to_u16(torch.from_numpy(cdf))
    assert np.array_equal(np.asarray(rows, dtype=np.uint16), got_rows)
    want = ref.ac_encode(rows, sym.tolist())
    got = codec.encode_float_cdf(torch.from_numpy(cdf), torch.from_numpy(sym), check_input_bounds=True)
    assert got == want
    assert ref.ac_decode(rows, want) == sym.tolist()
    assert np.array_equal(codec.decode_float_cdf(torch.from_numpy(cdf), got).numpy(), sym)


def test_skewed_stream_exercises_underflow_and_long_pending():
    # probabilities straddling 1/2 keep the interval around the midpoint -> many pending bits
    n = 3000
    cdf = np.tile(np.array([[0.0, 0.5 - 2e-5, 1.0]], np.float32), (n, 1))
    sym = (np.arange(n) % 2).astype(np.int16)
    rows = [ref.float_cdf_to_int(r) for r in cdf]
    want = ref.ac_encode(rows, sym.tolist())
    got = codec.encode_float_cdf(torch.from_numpy(cdf), torch.from_numpy(sym))
    assert got == want and ref.ac_decode(rows, want) == sym.tolist()
    assert np.array_equal(codec.decode_float_cdf(torch.from_numpy(cdf), got).numpy(), sym)


def test_empty_and_bounds():
    cdf = torch.zeros(0, 5)
    assert codec.decode_float_cdf(cdf, codec.encode_float_cdf(cdf, torch.zeros(0, dtype=torch.int16))).numel() == 0
    bad = torch.tensor([[0.0, 0.5, 1.2]])
    with pytest.raises(ValueError):
        codec.encode_float_cdf(bad, torch.tensor([0], dtype=torch.int16), check_input_bounds=True)
    with pytest.raises(ValueError):
        codec.encode_float_cdf(torch.tensor([[0.0, 0.5, 1.0]]), torch.tensor([2], dtype=torch.int16), check_input_bounds=True)
    with pytest.raises(RuntimeError):     # out-of-range symbol without the check: the C-ABI reports it
        codec.encode_float_cdf(torch.tensor([[0.0, 0.5, 1.0]]), torch.tensor([5], dtype=torch.int16))


def test_bernoulli_mask_stream_roundtrip(tmp_path):
    rng = np.random.default_rng(5)
    x = torch.from_numpy(np.where(rng.random(20000) < 0.7, 1.0, -1.0).astype(np.float32))
    p = torch.full_like(x, float((x > 0).float().mean()))
    f = str(tmp_path / "masks.b")
    bits = codec.encoder(x, p, f)
    assert torch.equal(codec.decoder(p, f), x)
    ent = 20000 * (-(0.7 * np.log2(0.7) + 0.3 * np.log2(0.3)))
    assert abs(bits - ent) < 0.02 * ent
    # same bytes as the oracle on the constant integer row
    row = ref.float_cdf_to_int([0.0, 1 - float(p[0]), 1.0])
    sym = ((x + 1) / 2).to(torch.int64).tolist()
    assert open(f, "rb").read() == ref.ac_encode([row] * len(sym), sym)


def test_rans_matches_oracle_and_roundtrips_with_escapes():
    from contextgs_amd.entropy_bottleneck import EntropyBottleneck
    torch.manual_seed(0)
    eb = EntropyBottleneck(12)
    eb.update()
    cdf, cl, off = eb._quantized_cdf.numpy(), eb._cdf_length.numpy(), eb._offset.numpy()
    assert (cdf[np.arange(12), cl - 1] == 65536).all() and (cdf[:, 0] == 0).all()
    rng = np.random.default_rng(1)
    sym = np.round(rng.normal(0, 4, size=(12, 257))).astype(np.int32)
    sym[3, 5], sym[2, 7], sym[0, 0], sym[11, 256] = 57, -33, 10, -11          # escapes on both sides and the edges
    data = codec.rans_encode_channels(sym, cdf, cl, off)
    assert data == ref.rans_encode(sym.tolist(), cdf.tolist(), cl.tolist(), off.tolist())
    back = codec.rans_decode_channels(data, 12, 257, cdf, cl, off)
    assert np.array_equal(back, sym)


def test_entropy_bottleneck_density_and_codec(host_density):
    from contextgs_amd.entropy_bottleneck import EntropyBottleneck, pmf_to_quantized_cdf
    torch.manual_seed(3)
    eb = EntropyBottleneck(12)
    with torch.no_grad():
        for m in eb.matrices:
            m.add_(0.1 * torch.randn_like(m))
    # likelihoods over the integers sum to ~1 per channel (a proper pmf)
    grid = torch.arange(-1500, 1501, dtype=torch.float32)[None, None, :].repeat(12, 1, 1)
    lik = eb._likelihood(grid)[:, 0, :]
    assert torch.allclose(lik.sum(1), torch.ones(12), atol=2e-3)
    # cumulative logits are monotone
    lg = eb._logits_cumulative(torch.linspace(-30, 30, 400)[None, None, :].repeat(12, 1, 1))
    assert (lg[:, 0, 1:] >= lg[:, 0, :-1]).all()
    eb.eval()
    x = torch.randn(500, 12) * 3
    xh, l = eb(x, training=False)
    assert torch.equal(xh, torch.round(x)) and (l > 0).all() and (l <= 1).all()
    xn, _ = eb(x, training=True)
    assert (xn - x).abs().max() <= 0.5
    q = pmf_to_quantized_cdf(np.array([0.5, 0.25, 0.25, 1e-12]))
    assert q[-1] == 65536 and (np.diff(q) >= 1).all()
    s = eb.compress(x.t().unsqueeze(0))
    y = eb.decompress(s, [500])
    assert torch.equal(y[0].t(), torch.round(x))
    assert eb.quantize(x, "symbols", eb._get_medians()[:, 0, 0]).dtype == torch.int32


def test_entropy_bottleneck_chunk_forms_equal_the_per_chunk_loop():
    """compress_chunks / decompress_chunks (host threads over the container's independent hyper strings) give the
    same bytes / values as a loop of compress() / decompress() (scene/gaussian_model.py:1082-1098, 1326-1336)."""
    from contextgs_amd.entropy_bottleneck import EntropyBottleneck
    torch.manual_seed(5)
    eb = EntropyBottleneck(12)
    eb.eval()
    x = torch.randn(2350, 12) * 4                       # ragged last chunk; a few escapes
    x[7, 3] = 4000.0
    loop = [eb.compress(x[s:s + 1000].t().unsqueeze(0))[0] for s in range(0, 2350, 1000)]
    assert eb.compress_chunks(x.t(), 1000) == loop
    sizes = [1000, 1000, 350]
    y = eb.decompress_chunks(loop, sizes)
    ref_y = torch.cat([eb.decompress([b], [n])[0] for b, n in zip(loop, sizes)], dim=1)
    assert torch.equal(y, ref_y) and torch.equal(y.t(), torch.round(x))
    assert eb.compress_chunks(x[:0].t(), 1000) == [] and eb.decompress_chunks([], []).shape == (12, 0)


@pytest.mark.parametrize("p1,n,seed", [(0.7, 5000, 1), (0.999, 20000, 2), (0.002, 20000, 3), (0.5, 3, 4), (0.35, 1, 5)])
def test_binary_mask_decoder_equals_generic_table_decoder(p1, n, seed):
    """The dedicated two-symbol loop behind the mask stream (no division / search, 64-bit bit buffer) returns the
    symbols of the generic table decoder and of the pure-Python oracle, incl. heavily skewed rows (long underflow
    runs), tiny streams, and streams cut short (zeros past the end)."""
    rng = np.random.default_rng(seed)
    sym = (rng.random(n) < p1).astype(np.int16)
    data = codec.bernoulli_encode_host(sym, p1)
    row = ref.float_cdf_to_int([0.0, 1 - float(np.float32(p1)), 1.0])
    assert data == ref.ac_encode([row] * n, sym.tolist())
    assert np.array_equal(codec.bernoulli_decode_host(data, n, p1), sym)
    table = torch.tensor([[0.0, 1 - float(np.float32(p1)), 1.0]], dtype=torch.float32).repeat(n, 1)
    for cut in (len(data), max(len(data) - 3, 0), len(data) // 2):
        fast = codec.bernoulli_decode_host(data[:cut], n, p1)
        generic = codec.decode_float_cdf(table, data[:cut]).numpy()
        assert np.array_equal(fast, generic)


def test_hyper_chunks_decode_straight_into_rows():
    """decompress_chunks_rows (chunk jobs writing dequantised rows of one [N, C] buffer) == decompress_chunks(...).t()."""
    from contextgs_amd.entropy_bottleneck import EntropyBottleneck
    torch.manual_seed(3)
    eb = EntropyBottleneck(12)
    with torch.no_grad():
        eb.quantiles[:, 0, 1] += torch.randn(12) * 0.3            # non-integer medians: the dequantisation matters
    eb.update(force=True)
    x = torch.randn(4321, 12) * 5
    x[17, 3], x[4000, 0] = 80.0, -75.0                            # escapes
    sizes = [1000, 1000, 1000, 1000, 321]
    strings = eb.compress_chunks(x.t(), 1000)
    want = eb.decompress_chunks(strings, sizes).t().contiguous()
    for tasks in (1, 3, 16):
        got = eb.decompress_chunks_rows(strings, sizes, tasks=tasks)()
        assert got.shape == want.shape and torch.equal(got, want)
    assert eb.decompress_chunks_rows([], [])().shape == (0, 12)
