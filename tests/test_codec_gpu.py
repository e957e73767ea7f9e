"""GPU parity of the device Gaussian codec and the bitstream container.

Bit-exact requirements (north_star: "decoded bitstreams are bit-exact"):
  * encode -> decode returns exactly the quantised values that were encoded;
  * the device coder emits exactly the bytes the table-driven coder (itself bit-exact vs the
    pure-Python oracle, tests/test_codec.py) emits for the device's own integer CDF table;
  * that integer table agrees with the oracle's fp32 restatement of the reference's table
    (utils/encodings.py:88-97) to within one count (erf implementations differ in the last ulp).
"""
import os

import numpy as np
import pytest
import torch

from oracle import codec_ref as ref

pytestmark = pytest.mark.gpu


def _params(n, seed, width=3.0):
    rng = np.random.default_rng(seed)
    mean = rng.normal(0, 2, n).astype(np.float32)
    scale = np.exp(rng.normal(0, 0.7, n)).astype(np.float32) * 0.8
    Q = (1 + np.tanh(rng.normal(0, 0.5, n))).clip(1e-3).astype(np.float32)
    x = mean + scale * rng.normal(0, width, n).astype(np.float32)
    xq = (np.round(x / Q) * Q).astype(np.float32)
    return xq, mean, scale, Q


def test_stream_bytes_equal_table_coder_on_device_table():
    from contextgs_amd import codec
    xq, mean, scale, Q = _params(3000, 0)
    T = lambda a: torch.from_numpy(a).cuda()
    streams, mn, mx = codec.gaussian_encode_streams(T(xq), T(mean), T(scale), T(Q), [0, 3000])
    table = codec.gaussian_cdf_table(T(mean), T(scale), T(Q), mn[0], mx[0])
    sym = (np.round(xq / Q) - mn[0]).astype(np.int64)
    assert sym.min() == 0 and sym.max() == mx[0] - mn[0]
    want = ref.ac_encode(table.tolist(), sym.tolist())
    assert streams[0] == want
    # the device table vs the fp32 restatement of the reference's float table: within one count
    oracle_rows = np.array([ref.float_cdf_to_int(r) for r in ref.gaussian_table(mean, scale, Q, int(mn[0]), int(mx[0]))])
    d = np.abs(oracle_rows[:, :-1].astype(np.int64) - table[:, :-1].astype(np.int64))
    d = np.minimum(d, 65536 - d)
    assert d.max() <= 1 and (d > 0).mean() < 0.02


@pytest.mark.parametrize("edges", [[0, 1], [0, 5, 5, 1000, 4096, 4097, 20000], [0, 50000, 100000, 123457]])
def test_roundtrip_bit_exact_ragged_streams(edges):
    from contextgs_amd import codec
    n = edges[-1]
    xq, mean, scale, Q = _params(n, len(edges))
    T = lambda a: torch.from_numpy(a).cuda()
    streams, mn, mx = codec.gaussian_encode_streams(T(xq), T(mean), T(scale), T(Q), edges)
    assert len(streams) == len(edges) - 1
    back = codec.gaussian_decode_streams(T(mean), T(scale), T(Q), edges, mn, mx, streams)
    assert torch.equal(back, T(xq))
    # rate sanity: within 3 % + slack of the model's own ideal code length
    tab_bits = 0.0
    for s in range(len(edges) - 1):
        a, b = edges[s], edges[s + 1]
        if b - a == 0 or b - a > 5000:
            continue
        t = codec.gaussian_cdf_table(T(mean[a:b]), T(scale[a:b]), T(Q[a:b]), mn[s], mx[s]).astype(np.int64)
        sym = (np.round(xq[a:b] / Q[a:b]) - mn[s]).astype(np.int64)
        hi = np.where(sym == t.shape[1] - 2, 65536, t[np.arange(b - a), np.minimum(sym + 1, t.shape[1] - 1)])
        ideal = -np.log2((hi - t[np.arange(b - a), sym]) / 65536.0).sum()
        assert len(streams[s]) * 8 <= ideal * 1.03 + 64


def test_row_broadcast_Q_and_reference_signatures(tmp_path):
    from contextgs_amd import codec
    from contextgs_amd.encodings import decoder_gaussian, encoder_gaussian
    rng = np.random.default_rng(7)
    n, c = 700, 50
    mean = rng.normal(0, 2, (n, c)).astype(np.float32)
    scale = np.exp(rng.normal(0, 0.5, (n, c))).astype(np.float32)
    Qrow = (1 + np.tanh(rng.normal(0, 0.5, (n,)))).clip(1e-3).astype(np.float32)
    x = (np.round((mean + scale * rng.normal(0, 2, (n, c))) / Qrow[:, None]) * Qrow[:, None]).astype(np.float32)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rows = torch.tensor([0, 250, 700]) * c
    s1, mn1, mx1 = codec.gaussian_encode_streams(T(x), T(mean), T(scale), T(Qrow), rows, q_div=c)
    Qfull = np.repeat(Qrow[:, None], c, 1)
    s2, mn2, mx2 = codec.gaussian_encode_streams(T(x), T(mean), T(scale), T(Qfull), rows, q_div=1)
    assert s1 == s2 and np.array_equal(mn1, mn2)
    back = codec.gaussian_decode_streams(T(mean), T(scale), T(Qrow), rows, mn1, mx1, s1, q_div=c)
    assert torch.equal(back.view(n, c), T(x))
    # reference-shaped single-stream API, via file and via bstream
    xf, mf, sf, qf = T(x).view(-1), T(mean).view(-1), T(scale).view(-1), T(Qfull).view(-1)
    f = str(tmp_path / "feat.b")
    b, bits, mn, mx = encoder_gaussian(xf, mf, sf, qf, file_name=f)
    assert bits == len(b) * 8 == os.path.getsize(f) * 8
    d1 = decoder_gaussian(mf, sf, qf, file_name=f, min_value=int(mn.item()), max_value=int(mx.item()))
    d2 = decoder_gaussian(mf, sf, qf, bstream=b, min_value=mn.cpu().int().item(), max_value=mx.cpu().int().item())
    assert torch.equal(d1, xf) and torch.equal(d2, xf)
    # scalar Q
    xs = torch.round(T(x).view(-1) / 0.5) * 0.5
    b, bits, mn, mx = encoder_gaussian(xs, mf, sf, 0.5)
    assert torch.equal(decoder_gaussian(mf, sf, 0.5, bstream=b, min_value=int(mn.item()), max_value=int(mx.item())), xs)


def _split_block(blob):
    """A version-2 block -> its 64 lane streams (header of 64 little-endian uint16 lengths, then the streams)."""
    lens = np.frombuffer(blob[:128], dtype="<u2").astype(np.int64)
    assert 128 + int(lens.sum()) == len(blob)
    ends = 128 + np.cumsum(lens)
    return [bytes(blob[e - n:e]) for e, n in zip(ends, lens)]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 3000, 64 * 40 + 17])
def test_lane_streams_equal_the_table_coder_on_device_table(n):
    """Container version 2 (cgs_gaussian_ac_encode_lanes): lane stream l of a block is byte for byte what the table coder
    (bit-exact vs the pure-Python oracle) emits for the block's symbols l, l + 64, ... on the device's own integer CDF rows;
    lanes without a symbol hold the empty stream's closing bits."""
    from contextgs_amd import codec
    xq, mean, scale, Q = _params(n, 40 + n)
    T = lambda a: torch.from_numpy(a).cuda()
    blob, lens, mn, mx = codec.gaussian_encode_packed(T(xq), T(mean), T(scale), T(Q), [0, n], lanes=True)
    assert lens.tolist() == [len(blob)]
    table = codec.gaussian_cdf_table(T(mean), T(scale), T(Q), mn[0], mx[0])
    sym = (np.round(xq / Q) - mn[0]).astype(np.int64)
    got = _split_block(blob.tobytes())
    for l in range(64):
        want = ref.ac_encode(table[l::64].tolist(), sym[l::64].tolist())
        assert got[l] == want, l
    back = codec.gaussian_decode_packed(T(mean), T(scale), T(Q), [0, n], mn, mx, blob, lens, lanes=True)
    assert torch.equal(back, T(xq))


@pytest.mark.parametrize("edges,width", [([0, 1], 3.0), ([0, 5, 5, 1000, 4096, 4097, 20000], 3.0),
                                         ([0, 32768, 65536, 100000], 3.0), ([0, 40000], 40.0), ([0, 70000, 70001], 0.05)])
def test_lane_blocks_roundtrip_bit_exact(edges, width):
    """Ragged blocks incl. empty ones, symbols far in the tails (width 40: the decoder's inverse-CDF guess is clamped and the
    walk takes over) and nearly deterministic ones (width 0.05); rate within 3 % + the per-lane termination of the ideal."""
    from contextgs_amd import codec
    n = edges[-1]
    xq, mean, scale, Q = _params(n, len(edges), width)
    T = lambda a: torch.from_numpy(a).cuda()
    blob, lens, mn, mx = codec.gaussian_encode_packed(T(xq), T(mean), T(scale), T(Q), edges, lanes=True)
    assert len(lens) == len(edges) - 1 and int(lens.sum()) == len(blob)
    back = codec.gaussian_decode_packed(T(mean), T(scale), T(Q), edges, mn, mx, blob, lens, lanes=True)
    assert torch.equal(back, T(xq))
    # same symbols through the wave-per-stream coder: the lane layout costs the header + 64 terminations per block
    blob1, lens1, mn1, mx1 = codec.gaussian_encode_packed(T(xq), T(mean), T(scale), T(Q), edges)
    assert np.array_equal(mn, mn1) and np.array_equal(mx, mx1)
    for a, b in zip(lens, lens1):
        assert int(a) <= int(b) + 128 + 64 * 3 + 8
    # grouped form (what the container driver calls)
    g = (T(xq), T(mean), T(scale), T(Q), edges, 1)
    (blob2, lens2, mn2, mx2), = codec.gaussian_encode_groups([g], lanes=True)
    assert blob2.tobytes() == blob.tobytes()
    out, = codec.gaussian_decode_groups([(g[1], g[2], g[3], edges, mn2, mx2, blob2, lens2, 1)], lanes=True)
    assert torch.equal(out, T(xq))


@pytest.mark.parametrize("N,block", [(1, 64), (65, 64), (5000, 1024), (40017, 32768)])
def test_table_lane_blocks_hyper_roundtrip_and_oracle_bytes(N, block):
    """Container version 2's hyper.b (EntropyBottleneck.compress_lanes / decompress_lanes_rows): the per-channel integer tables
    of update(), blocks of `block` anchors of one channel, 64 interleaved lane streams; values outside a table's support take
    the escape slot + sign / unary length / low bits as equiprobable binary symbols.  Every lane stream equals the bit-list
    oracle's stream for that symbol sequence; decoding returns round(x - median) + median."""
    from contextgs_amd.entropy_bottleneck import EntropyBottleneck
    torch.manual_seed(N)
    eb = EntropyBottleneck(12).cuda()
    eb.update(force=True)
    x = (torch.randn(12, N, device="cuda") * 6.0)
    x[0, 0] = 250.0                       # far outside: a long escape
    if N > 3:
        x[3, 3] = -97.5
    blob, lens = eb.compress_lanes(x, block)
    rows = eb.decompress_lanes_rows(blob, lens, N, block)
    med = eb._get_medians()[:, 0, 0]
    want = torch.round(x - med[:, None]) + med[:, None]
    assert torch.equal(rows.t(), want)
    # oracle bytes of the first and the last block
    sym = eb.quantize(x, "symbols", eb._get_medians()[:, 0]).cpu().numpy()
    cdf, cl, of = eb._quantized_cdf.cpu().numpy(), eb._cdf_length.cpu().numpy(), eb._offset.cpu().numpy()
    nper = -(-N // block)
    pos = np.concatenate([[0], np.cumsum(lens)])
    BIT = [0, 32768, 65536]
    for blk in sorted({0, len(lens) - 1}):
        c, k = blk // nper, blk % nper
        seq = sym[c, k * block:(k + 1) * block]
        table = cdf[c, :cl[c]].tolist()
        max_value = int(cl[c]) - 2
        got = _split_block(blob[pos[blk]:pos[blk + 1]].tobytes())
        for l in range(64):
            R, S = [], []
            for v in seq[l::64]:
                raw = int(v) - int(of[c])
                esc = raw < 0 or raw >= max_value
                R.append(table); S.append(max_value if esc else raw)
                if esc:
                    m = -raw if raw < 0 else raw - max_value + 1
                    nb = m.bit_length() - 1
                    bits = [1 if raw < 0 else 0] + [0] * nb + [1] + [(m >> j) & 1 for j in range(nb - 1, -1, -1)]
                    R += [BIT] * len(bits); S += bits
            assert got[l] == ref.ac_encode(R, S), (blk, l)


@pytest.mark.parametrize("p0", [0.03, 0.31, 0.5, 0.97])
def test_bernoulli_chunk_streams_equal_the_host_coder(p0):
    """Container version 2's mask streams (cgs_bernoulli_ac_encode / _decode, one wave per chunk stream): every stream is
    byte for byte the host coder's stream for its symbols (which tests/test_codec.py pins against the bit-list oracle),
    incl. empty, 1-symbol, 63/64/65-symbol and ragged streams; decode is the inverse; non-binary input is refused."""
    from contextgs_amd import codec
    rng = np.random.default_rng(int(p0 * 1000))
    lens = [0, 1, 63, 64, 65, 1000, 0, 10_000, 4097, 2]
    edges = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    sym = (rng.random(int(edges[-1])) < p0).astype(np.float32)
    sym_d = torch.from_numpy(sym).cuda()
    blob, blens = codec.bernoulli_encode_packed(sym_d, p0, edges)
    want = [codec.bernoulli_encode_host(sym[a:b].astype(np.int16), p0) for a, b in zip(edges[:-1], edges[1:])]
    assert blens.tolist() == [len(w) for w in want]
    assert blob.tobytes() == b"".join(want)
    out = codec.bernoulli_decode_packed(p0, edges, blob, blens)
    assert torch.equal(out, sym_d)
    # a side-stream job gives the same bytes
    job = codec.BernoulliEncodeJob(sym_d, p0, edges, torch.cuda.Stream())
    blob2, blens2 = job.result()
    assert blob2.tobytes() == blob.tobytes() and np.array_equal(blens, blens2)
    bad = sym_d.clone(); bad[5000] = 0.5
    with pytest.raises(RuntimeError, match="outside"):
        codec.bernoulli_encode_packed(bad, p0, edges)


@pytest.mark.parametrize("N,seed,version", [(3000, 2, 1), (12000, 5, 1), (3000, 2, 2), (12000, 5, 2)])
def test_container_encode_decode_roundtrip(tmp_path, N, seed, version):
    """conduct_encoding -> files -> conduct_decoding on a second model object: every decoded
    attribute equals the encoder-side quantised value bit for bit; rendering the decoded model
    equals rendering the encoder's quantised model.  Both container versions (codec_driver.CONTAINER_VERSION)."""
    import golden_inputs as gi
    from contextgs_amd import context_model as cm
    from contextgs_amd.model import GaussianModel

    def build():
        pc = GaussianModel(voxel_size=0.01)
        sd = pc.state_dict()
        for k, v in gi.mlp_weights(seed).items():
            sd[k] = torch.from_numpy(v).cuda()
        pc.load_state_dict(sd, strict=False)
        st = gi.anchor_state(N, seed)
        pc.set_state(st["anchor"], st["offset"], st["mask"], st["feat"], st["hyper"], st["scaling"])
        pc.update_anchor_bound()
        return pc

    enc = build()
    enc.eval()
    d = str(tmp_path / "bitstreams")
    from contextgs_amd.codec_driver import conduct_encoding
    info = conduct_encoding(enc, d, container_version=version)
    assert "EncTime" in info and "Total" in info
    for f in ["anchor.npy", "hyper.b", "masks.b", "meta.b", "mlp.pt"] + [f"{a}{l}.b" for a in ("feat", "scaling", "offsets") for l in range(3)]:
        assert os.path.exists(os.path.join(d, f)), f
    est = enc.estimate_final_bits()
    assert "Estimated sizes" in est

    dec = build()
    with torch.no_grad():           # scramble: everything must come from the files
        dec._anchor_feat.zero_(); dec._offset.zero_(); dec._hyper_latent.zero_(); dec._scaling.zero_()
        for p in dec.mlp_grid.parameters():
            p.zero_()
    dec.eval()
    info = dec.conduct_decoding(d)
    assert "DecTime" in info and dec.decoded_version

    with torch.no_grad():
        m = enc.get_mask_anchor
        nv = int(m.sum())
        anchor = enc.get_anchor[m]
        f, s, o = cm.multi_scale_generating(enc, anchor, enc._hyper_latent[m], enc._anchor_feat[m], enc._offset[m],
                                            enc.get_scaling[m], enc.get_mask[m], None, predict_bpp=False, training=False)
        assert torch.equal(dec._anchor[:nv], anchor)
        assert torch.equal(dec._mask[:nv], enc.get_mask[m])
        assert torch.equal(dec._hyper_latent[:nv], torch.round(enc._hyper_latent[m]))
        assert torch.equal(dec._anchor_feat[:nv], f)
        assert torch.equal(dec._scaling[:nv], s)
        assert torch.equal(dec._offset[:nv], o * enc.get_mask[m])        # masked-out offsets are not transmitted
        assert float(dec._anchor_feat[nv:].abs().sum()) == 0.0
    # coded size vs the rate model's estimate: with random (untrained) grid MLPs many symbols sit below the
    # estimator's 1e-6 likelihood floor (19.9 bit) while the coder never spends more than 16 bit on one, so
    # the stream may be up to ~30 % smaller than the estimate but never much larger
    meta = torch.load(os.path.join(d, "meta.b"), weights_only=False)
    coded = sum(sum(v) for v in meta[10].values()) + sum(sum(v) for v in meta[11].values()) + sum(sum(v) for v in meta[12].values())
    sums = cm.multi_scale_generating(enc, anchor, enc._hyper_latent[m], enc._anchor_feat[m], enc._offset[m],
                                     enc.get_scaling[m], binary_grid_masks=enc.get_mask[m], predict_bpp=True,
                                     return_sum_bits=True)
    est_bits = sums[2] + sums[3] + sums[4]
    assert 0.65 * est_bits <= coded <= 1.08 * est_bits
    # the mask stream of the container (scene/gaussian_model.py:1265-1269; coded by the branch-free host loop on a pool
    # thread) is byte for byte the bit-list oracle's stream for the same symbols and probability
    prob = float(meta[8])
    sym = ((enc.get_mask[m].reshape(-1) > 0).to(torch.int64)).cpu().tolist()
    row = ref.float_cdf_to_int([0.0, float(np.float32(1.0) - np.float32(prob)), 1.0])
    masks_b = open(os.path.join(d, "masks.b"), "rb").read()
    if version == 1:
        assert len(meta) == 14
        assert masks_b == ref.ac_encode([row] * len(sym), sym)
    else:
        # version 2: the same symbols cut into 1000-anchor chunk streams, each the oracle's stream for its symbols
        assert len(meta) == 15 and meta[14]["version"] == 2
        from contextgs_amd.codec_driver import _level_blocks
        ck, K = meta[14]["chunk"], enc.n_offsets
        want = b"".join(ref.ac_encode([row] * len(sym[a:a + ck["masks"] * K]), sym[a:a + ck["masks"] * K])
                        for a in range(0, len(sym), ck["masks"] * K))
        assert masks_b == want and sum(meta[14]["bit_masks"]) == 8 * len(masks_b)
        # feat / scaling: ceil(symbols / block) blocks per level, each a 128-byte header + 64 lane streams
        for l, n_l in enumerate(reversed(meta[13])):
            B_of = lambda n_sym: _level_blocks(n_l, n_sym, enc.feat_dim, int(meta[14].get("block_policy", 0)), int(meta[14]["block_symbols"]))
            assert (len(meta[10][l]) == -(-n_l * enc.feat_dim // B_of(n_l * enc.feat_dim))
                    and len(meta[11][l]) == -(-n_l * 6 // B_of(n_l * 6)))
            blob = open(os.path.join(d, f"feat{l}.b"), "rb").read()
            assert len(blob) * 8 == sum(meta[10][l])
            pos = 0
            for bits in meta[10][l]:
                _split_block(blob[pos:pos + bits // 8])
                pos += bits // 8


@pytest.mark.parametrize("N", [3000, 10000])
def test_container_bytes_match_the_committed_digest(tmp_path, N):
    """Known-answer pin of the bitstream (tests/container_digest.py): re-encoding the golden model gives, byte for byte, the
    files whose sha256 / length were committed from the MI355X (tools/make_container_golden.py), for both container
    versions; so do the header's content and every decoded tensor."""
    import json
    from container_digest import container_digest
    path = os.path.join(os.path.dirname(__file__), "golden", f"container_n{N}.json")
    want = json.load(open(path))["containers"]
    for w in want:
        got = container_digest(w["N"], w["seed"], w["container_version"], tmp_path)
        for f, (sha, size) in w["files"].items():
            assert got["files"][f] == [sha, size], (w["container_version"], f, got["files"][f], [sha, size])
        assert got["meta"] == w["meta"], ("meta.b content", w["container_version"])
        assert got["decoded"] == w["decoded"], w["container_version"]


def test_grouped_launch_is_byte_identical_to_per_group_launches():
    """gaussian_encode_groups / gaussian_decode_groups (all streams of several attribute groups in ONE coder launch,
    row-broadcast Q expanded per symbol) give the bytes / values of separate gaussian_encode_streams calls, incl. an
    empty group and an empty stream; the packed blob is the b"".join of the chunk strings."""
    from contextgs_amd import codec
    dev = "cuda"
    specs = [(50, [0, 50 * 40, 50 * 100, 50 * 137]), (6, [0, 6 * 100, 6 * 100, 6 * 137]), (1, [0, 811, 2000]), (1, [0])]
    groups, singles = [], []
    for gi_, (q_div, edges) in enumerate(specs):
        n = edges[-1]
        xq, mean, scale, Qs = _params(n, 20 + gi_)
        rows = max(n // q_div, 0)
        Qrow = Qs[:rows].copy()
        Qfull = np.repeat(Qrow, q_div)
        xq = (np.round((mean + (xq - mean)) / np.maximum(Qfull, 1e-3)) * Qfull).astype(np.float32) if n else xq
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        g = (t(xq), t(mean), t(scale), t(Qrow), edges, q_div)
        groups.append(g)
        singles.append(codec.gaussian_encode_streams(*g[:5], q_div=q_div))
    packed = codec.gaussian_encode_groups(groups)
    dec_groups = []
    for g, (streams, mn, mx), (blob, lens, mn2, mx2) in zip(groups, singles, packed):
        assert blob.tobytes() == b"".join(streams) and lens.tolist() == [len(b) for b in streams]
        assert np.array_equal(mn, mn2) and np.array_equal(mx, mx2)
        dec_groups.append((g[1], g[2], g[3], g[4], mn2, mx2, blob, lens, g[5]))
    for g, out in zip(groups, codec.gaussian_decode_groups(dec_groups)):
        assert torch.equal(out, g[0])


def test_staged_files_hold_release_and_deferred_download(tmp_path):
    """The container's file plumbing: StagedFiles (files -> pinned -> device through cgs_pread_ranges, the files behind
    `start` held until release() / the first get()) and the encoder's deferred download (pieces picked up by write_file as
    they land, through cgs_pwrite_ranges)."""
    from contextgs_amd import codec
    rng = np.random.default_rng(11)
    sizes = [1000, 0, (40 << 20) + 13, 70_000]                 # an empty file, one larger than a 32 MB batch
    blobs = [rng.integers(0, 256, n, dtype=np.uint8) for n in sizes]
    paths = []
    for k, b in enumerate(blobs):
        p = str(tmp_path / f"f{k}.b")
        b.tofile(p)
        paths.append(p)
    dev = torch.device("cuda")
    for start in (None, 1, 0):
        st = codec.StagedFiles(paths, dev, start=start)
        if start == 1:
            assert torch.equal(st.get(paths[0]).cpu(), torch.from_numpy(blobs[0]))          # staged before the release
            assert not st.ready[paths[2]].is_set()
        for p, b in zip(reversed(paths), reversed(blobs)):                                   # get() of a held file releases
            got = st.get(p)
            torch.cuda.synchronize()
            assert got.numel() == b.size and torch.equal(got.cpu(), torch.from_numpy(b))
        st.wait_all()
    os.remove(paths[3])
    with pytest.raises(OSError):
        codec.StagedFiles(paths, dev)             # sizes are read at construction: a vanished file fails there
    # deferred download -> files
    n = 3_000_000
    g = torch.Generator(device="cuda").manual_seed(5)
    mean = torch.randn(n, device="cuda", generator=g)
    scale = torch.rand(n, device="cuda", generator=g) * 2 + 0.05
    Q = torch.ones(n, device="cuda")
    x = torch.round(mean + scale * torch.randn(n, device="cuda", generator=g))
    edges = torch.arange(0, n + 1, 32768).tolist()
    if edges[-1] != n:
        edges.append(n)
    ref = codec.gaussian_encode_packed(x, mean, scale, Q, torch.tensor(edges), lanes=True)
    blob, lens, mn, mx = codec.gaussian_encode_packed(x, mean, scale, Q, torch.tensor(edges), staging=True, lanes=True, deferred=True)
    ready = codec.stage_ready()
    assert ready is not None
    out = str(tmp_path / "coded.b")
    for j in codec.write_file(out, blob, piece=1 << 20, ready=ready):
        j.result()
    assert np.array_equal(np.fromfile(out, dtype=np.uint8), ref[0]) and np.array_equal(lens, ref[1])
    ready.wait_all()
    assert np.array_equal(blob, ref[0])


def test_a_corrupt_lane_block_is_reported_not_decoded(tmp_path):
    """ container version 2: the lane decoders read the 64 lane lengths of a block header bytewise and check them against
    the block length of meta.b — flipping a length byte in feat0.b makes conduct_decoding raise instead of steering device reads
    by the corrupt value; the untouched container still decodes afterwards (the status word is per decoding)."""
    import golden_inputs as gi
    from contextgs_amd.codec_driver import conduct_encoding
    from contextgs_amd.model import GaussianModel

    def build():
        pc = GaussianModel(voxel_size=0.01)
        sd = pc.state_dict()
        for k, v in gi.mlp_weights(2).items():
            sd[k] = torch.from_numpy(v).cuda()
        pc.load_state_dict(sd, strict=False)
        st = gi.anchor_state(3000, 2)
        pc.set_state(st["anchor"], st["offset"], st["mask"], st["feat"], st["hyper"], st["scaling"])
        pc.update_anchor_bound()
        pc.eval()
        return pc

    d = str(tmp_path / "bits")
    conduct_encoding(build(), d, container_version=2)
    f = os.path.join(d, "feat0.b")
    good = open(f, "rb").read()
    bad = bytearray(good)
    bad[0] ^= 0x40                       # lane 0's length, low byte
    open(f, "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="malformed"):
        build().conduct_decoding(d)
    open(f, "wb").write(good)
    build().conduct_decoding(d)
