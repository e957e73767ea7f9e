"""Entropy-coding front end over libcgs_hip.so's codec
entry points (csrc/codec.hip):

  * `encode_float_cdf` / `decode_float_cdf`        — the `torchac` API the reference
    imports (utils/encodings.py:6), host arithmetic coder on uint16 CDFs;
  * `encoder_gaussian` / `decoder_gaussian`         — utils/encodings.py:83-144, same
    signatures and return values, but the per-symbol integer CDF is evaluated inside
    the device coder (no [n_sym, L] float table, no PCIe round trip);
  * `gaussian_encode_packed` / `gaussian_decode_packed` (+ `_streams`, `_groups` forms) —
    the batched form the container driver uses: any number of 1000-anchor chunk streams,
    of any mix of levels / attributes, coded by ONE launch, one wave per stream, the
    byte streams packed back to back on the device (`_groups` also shards the streams
    over the ranks of a process group);
  * `encoder` / `decoder`                           — Bernoulli mask stream, :147-180;
  * `rans_encode_channels` / `rans_decode_channels` — hyper-prior symbols.

There is no CPU fallback for the Gaussian codec: tensors must live on the HIP device.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# ---- torchac-compatible host coder ---------------------------------------------------------
def _cdf_to_u16(cdf_float: torch.Tensor) -> np.ndarray:
    L = _lib.lib()
    cdf = np.ascontiguousarray(cdf_float.detach().cpu().numpy(), dtype=np.float32)
    Lp = cdf.shape[-1]
    flat = cdf.reshape(-1, Lp)
    out = np.empty(flat.shape, dtype=np.uint16)
    _lib.check(L.cgs_cdf_float_to_u16_host(_np_ptr(flat), flat.shape[0], Lp, _np_ptr(out)), "cgs_cdf_float_to_u16_host")
    return out


def encode_float_cdf(cdf_float, sym, needs_normalization=True, check_input_bounds=False) -> bytes:
    """torchac.encode_float_cdf: cdf_float [..., Lp] in [0,1], sym int16 [...] in [0, Lp-2]."""
    L = _lib.lib()
    if not needs_normalization:
        raise NotImplementedError("only the normalising conversion the reference uses is implemented")
    if check_input_bounds:
        if cdf_float.min() < 0:
            raise ValueError(f"cdf_float.min() == {cdf_float.min()}, should be >=0.!")
        if cdf_float.max() > 1:
            raise ValueError(f"cdf_float.max() == {cdf_float.max()}, should be <=1.!")
        if sym.max() >= cdf_float.shape[-1] - 1:
            raise ValueError("symbol out of range for the given CDF")
    cdf = _cdf_to_u16(cdf_float)
    s = np.ascontiguousarray(sym.detach().cpu().numpy().reshape(-1), dtype=np.int16)
    assert s.shape[0] == cdf.shape[0], "one CDF row per symbol"
    cap = L.cgs_ac_max_bytes(s.shape[0])
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    _lib.check(L.cgs_ac_encode_table_host(_np_ptr(cdf), cdf.shape[1], _np_ptr(s), s.shape[0], _np_ptr(out), cap,
                                          C.byref(n)), "cgs_ac_encode_table_host")
    return out[: n.value].tobytes()


def decode_float_cdf(cdf_float, byte_stream: bytes, needs_normalization=True) -> torch.Tensor:
    """torchac.decode_float_cdf -> int16 symbols shaped like cdf_float[..., 0]."""
    L = _lib.lib()
    cdf = _cdf_to_u16(cdf_float)
    buf = np.frombuffer(byte_stream, dtype=np.uint8)
    out = np.empty(cdf.shape[0], dtype=np.int16)
    _lib.check(L.cgs_ac_decode_table_host(_np_ptr(cdf), cdf.shape[1], cdf.shape[0], _np_ptr(buf) if buf.size else None,
                                          buf.size, _np_ptr(out)), "cgs_ac_decode_table_host")
    return torch.from_numpy(out).reshape(cdf_float.shape[:-1])


# ---- Bernoulli mask stream (utils/encodings.py:147-180) ---------------------------------------
def _bernoulli_row(p: float) -> np.ndarray:
    row = np.array([[0.0, 1.0 - p, 1.0]], dtype=np.float32)
    out = np.empty((1, 3), dtype=np.uint16)
    _lib.check(_lib.lib().cgs_cdf_float_to_u16_host(_np_ptr(row), 1, 3, _np_ptr(out)), "cgs_cdf_float_to_u16_host")
    return out[0]


_POOL = None


def host_pool():
    """Host threads for the format's serial streams (mask stream, hyper rANS strings): the coders are C calls
    through ctypes, which release the GIL, so independent streams run concurrently with each other and with the
    device launches of the level loop."""
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(2, min(32, os.cpu_count() or 2)), thread_name_prefix="cgs-codec")
    return _POOL


def write_file(path: str, blob, piece: int = 8 << 20, ready=None):
    """Write a uint8 ndarray / bytes to `path` on the pool: files above `piece` bytes are written as concurrent os.pwrite
    slices (one thread copies ~9 GB/s into the page cache; the finest level's feature file is ~80 MB at 1 M anchors and
    sat on the encoder's tail for 10 ms).  Returns the futures to wait on.
    ready: a StageReady whose download `blob` may still be part of — every slice waits for its own bytes only."""
    import os
    mv = memoryview(blob).cast("B") if not isinstance(blob, (bytes, bytearray)) else memoryview(blob)
    n = len(mv)
    arrived = (lambda lo, hi: ready.wait(blob, lo, hi)) if ready is not None and isinstance(blob, np.ndarray) else (lambda lo, hi: None)
    if n <= piece:
        def small():
            arrived(0, n)
            with open(path, "wb") as f:
                f.write(mv)
        return [host_pool().submit(small)]
    # nothing on the caller's thread touches the file: truncating an existing 80 MB file (the previous container in the same
    # directory) alone takes ~10 ms.  One pool job per file; the slices themselves are written by C++ workers
    # (cgs_pwrite_ranges: no interpreter thread per slice queueing for the GIL), in batches that follow the download.
    if isinstance(blob, np.ndarray):
        base = blob.ctypes.data

        def big():
            for b0 in range(0, n, 4 * piece):
                b1 = min(n, b0 + 4 * piece)
                arrived(b0, b1)
                file_ranges(True, [(path, lo, min(b1, lo + piece) - lo, base + lo) for lo in range(b0, b1, piece)], 8)
            os.truncate(path, n)
        return [host_pool().submit(big)]

    def part(lo):
        fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
        try:
            hi = min(n, lo + piece)
            while lo < hi:
                lo += os.pwrite(fd, mv[lo:hi], lo)
        finally:
            os.close(fd)
    jobs = [host_pool().submit(part, lo) for lo in range(0, n, piece)]

    def finish():
        for j in jobs:
            j.result()
        os.truncate(path, n)
    return [host_pool().submit(finish)]


def bernoulli_encode_host(sym: np.ndarray, p0: float) -> bytes:
    """sym int16 [n] in {0,1}, P(1) = p0 -> the single arithmetic-coded stream of utils/encodings.py:147-163."""
    L = _lib.lib()
    row = _bernoulli_row(np.float32(p0))
    sym = np.ascontiguousarray(sym, dtype=np.int16)
    cap = L.cgs_ac_max_bytes(sym.shape[0])
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    _lib.check(L.cgs_ac_encode_const_host(_np_ptr(row), 3, _np_ptr(sym), sym.shape[0], _np_ptr(out), cap, C.byref(n)),
               "cgs_ac_encode_const_host")
    return out[: n.value].tobytes()


def bernoulli_decode_host(data, n: int, p0: float) -> np.ndarray:
    """inverse of bernoulli_encode_host -> int16 [n] in {0,1} (utils/encodings.py:166-180)."""
    L = _lib.lib()
    row = _bernoulli_row(np.float32(p0))
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    out = np.empty(int(n), dtype=np.int16)
    _lib.check(L.cgs_ac_decode_const_host(_np_ptr(row), 3, out.shape[0], _np_ptr(buf) if buf.size else None, buf.size,
                                          _np_ptr(out)), "cgs_ac_decode_const_host")
    return out


def encoder(x, p, file_name):
    """x in {-1,+1}, p = P(+1) per element (the reference passes one global value).  Returns bit length."""
    assert file_name[-2:] == ".b"
    p = p.detach().reshape(-1)
    p0 = float(p[0].item()) if p.numel() else 0.5
    sym = torch.floor((x.detach().reshape(-1) + 1) / 2).to(torch.int16).cpu().numpy()
    data = bernoulli_encode_host(sym, p0)
    with open(file_name, "wb") as f:
        f.write(data)
    return len(data) * 8


def decoder(p, file_name):
    assert file_name[-2:] == ".b"
    dvc = p.device
    pf = p.detach().reshape(-1)
    with open(file_name, "rb") as f:
        data = f.read()
    out = bernoulli_decode_host(data, pf.numel(), float(pf[0].item()) if pf.numel() else 0.5)
    return (torch.from_numpy(out).to(torch.float32) * 2 - 1).to(dvc)


# ---- Bernoulli chunk streams on the device (container version 2) -------------------------------------
def bernoulli_c1(p0: float) -> int:
    """The one interior entry of the mask stream's integer CDF row [0, c1, 2^16] (P(1) = p0), through the same float ->
    uint16 conversion the host coder uses (utils/encodings.py:151-157)."""
    return int(_bernoulli_row(np.float32(p0))[1])


class BernoulliEncodeJob:
    """Mask chunk streams coded by ONE device launch (one wave per stream) on a side stream; result() downloads
    (packed bytes, per-stream byte lengths).  Stream s holds exactly bernoulli_encode_host(sym[off[s]:off[s+1]], p0)."""

    def __init__(self, sym01: torch.Tensor, p0: float, stream_off, side_stream=None):
        L = _lib.lib()
        _lib.require_device(sym01)
        self.dev = dev = sym01.device
        self.sym = sym = _f(sym01).reshape(-1)
        off_h = torch.as_tensor(stream_off, dtype=torch.int64).cpu()
        self.S = S = int(off_h.numel()) - 1
        self.stream = side_stream or torch.cuda.current_stream(dev)
        if S <= 0:
            return
        assert int(off_h[-1]) == sym.numel() and int(off_h[0]) == 0
        caps = ((off_h[1:] - off_h[:-1]) * 2 + 16 + 7) // 8 * 8
        out_off_h = torch.zeros(S + 1, dtype=torch.int64)
        out_off_h[1:] = torch.cumsum(caps, 0)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            off_d, out_off = off_h.to(dev, non_blocking=True), out_off_h.to(dev, non_blocking=True)
            out = torch.empty(int(out_off_h[-1]) + 16, dtype=torch.uint8, device=dev)
            self.st_len = st_len = torch.zeros(S + 1, dtype=torch.int32, device=dev)
            raw = self.stream.cuda_stream
            _lib.check(L.cgs_bernoulli_ac_encode(_lib.ptr(sym), bernoulli_c1(p0), _lib.ptr(off_d), S, _lib.ptr(out),
                                                 _lib.ptr(out_off), _lib.ptr(st_len[1:]), _lib.ptr(st_len[:1]), raw),
                       "cgs_bernoulli_ac_encode")
            dst_off = torch.zeros(S + 1, dtype=torch.int64, device=dev)
            torch.cumsum(st_len[1:], 0, out=dst_off[1:])
            # an upper bound sizes the packed buffer, so nothing is read back before the compaction is enqueued
            self.packed = packed = torch.empty(int(out_off_h[-1]) + 16, dtype=torch.uint8, device=dev)
            _lib.check(L.cgs_streams_compact(_lib.ptr(out), _lib.ptr(out_off), _lib.ptr(st_len[1:]), _lib.ptr(dst_off), S,
                                             _lib.ptr(packed), raw), "cgs_streams_compact")
            for t in (sym, off_d, out_off, out, st_len, dst_off, packed):
                t.record_stream(self.stream)

    def result(self):
        if self.S <= 0:
            return np.zeros(0, np.uint8), np.zeros(0, np.int64)
        with torch.cuda.stream(self.stream):
            st_len_h = self.st_len.cpu().numpy()
            if int(st_len_h[0]) != 0:
                raise RuntimeError("bernoulli codec: " + ("symbol outside {0, 1}" if int(st_len_h[0]) == 1
                                                          else "stream overflowed its buffer"))
            lens = st_len_h[1:].astype(np.int64)
            nbytes = int(lens.sum())
            blob = self.packed[:nbytes].cpu().numpy()
        return blob, lens


def bernoulli_encode_packed(sym01, p0, stream_off):
    """sym01 flat float {0,1} device tensor -> (blob uint8 ndarray of the S streams back to back, lens int64 [S])."""
    return BernoulliEncodeJob(sym01, p0, stream_off).result()


def bernoulli_decode_packed(p0, stream_off, blob, lens, device=None):
    """Inverse of bernoulli_encode_packed -> flat float32 {0,1} device tensor.  blob: bytes / uint8 ndarray / uint8 device
    tensor (followed by >= 16 readable bytes of its storage, as StagedFiles hands them out)."""
    L = _lib.lib()
    off_h = torch.as_tensor(stream_off, dtype=torch.int64).cpu()
    S = int(off_h.numel()) - 1
    if isinstance(blob, torch.Tensor) and blob.is_cuda:
        dev, in_d = blob.device, blob
    else:
        dev = torch.device(device if device is not None else "cuda")
        buf = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
        in_d = torch.zeros(buf.size + 16, dtype=torch.uint8, device=dev)
        if buf.size:
            in_d[: buf.size].copy_(torch.from_numpy(np.ascontiguousarray(buf) if buf.flags.writeable else buf.copy()))
    n = int(off_h[-1]) if S >= 0 and off_h.numel() else 0
    out = torch.empty(n, dtype=torch.float32, device=dev)
    if S <= 0:
        return out
    lens = np.asarray(lens, dtype=np.int64)
    assert lens.shape[0] == S
    in_off_h = np.zeros(S + 1, dtype=np.int64)
    np.cumsum(lens, out=in_off_h[1:])
    assert int(in_off_h[-1]) <= int(in_d.numel()), "stream lengths exceed the blob"
    in_off, off_d = _upload_small([in_off_h, off_h], dev)
    _lib.check(L.cgs_bernoulli_ac_decode(bernoulli_c1(p0), _lib.ptr(off_d), S, _lib.ptr(in_d), _lib.ptr(in_off), _lib.ptr(out),
                                         _lib.current_stream()), "cgs_bernoulli_ac_decode")
    return out


# ---- batched device Gaussian codec ---------------------------------------------------------------
def _f(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


_STAGE = {}


def _pinned_staging(nbytes: int, key: str = "bitstream") -> torch.Tensor:
    """Grow-only pinned host buffers, one per purpose (allocating pinned memory costs milliseconds per call, and a buffer a
    host thread is still reading must not be reused for something else)."""
    buf = _STAGE.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _STAGE[key] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
    return buf


_STAGE_PIECE = 8 << 20
_STAGE_READY = None


class StageReady:
    """Arrival of a staged download, piece by piece (gaussian_encode_packed(staging=True))."""

    def __init__(self, base, nbytes, events, keep):
        self.base, self.nbytes, self.events, self._keep = base, nbytes, events, keep     # keep: the device source of the copies

    def wait_all(self):
        for ev in self.events:
            ev.synchronize()
        self._keep = None

    def wait(self, arr, lo=0, hi=None):
        """Block until bytes [lo, hi) of `arr` — a uint8 view into the staging buffer — have landed (no-op for other arrays)."""
        a0 = arr.__array_interface__["data"][0] if isinstance(arr, np.ndarray) else None
        if a0 is None or not (self.base <= a0 < self.base + self.nbytes):
            return
        hi = arr.nbytes if hi is None else hi
        if hi <= lo:
            return
        first, last = (a0 - self.base + lo) // _STAGE_PIECE, (a0 - self.base + hi - 1) // _STAGE_PIECE
        for i in range(first, min(last, len(self.events) - 1) + 1):
            self.events[i].synchronize()


def stage_ready():
    """The arrival tracker of the last staged download of this process, or None when it was waited for inline."""
    return _STAGE_READY


def to_host_pinned(t: torch.Tensor, key: str) -> np.ndarray:
    """Device tensor -> numpy array backed by the reused pinned buffer `key` (valid until the next call with that key):
    a pinned download runs at PCIe rate, a pageable `.cpu()` at a fifth of it and on a runtime staging thread."""
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    stage = _pinned_staging(nbytes, key)[:nbytes].view(t.dtype).view(t.shape)
    stage.copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return stage.numpy()


_SMALL_RING = {"slots": [], "next": 0}


_KEYED_SLOTS = {}


def _upload_small(arrays, dev, key=None):
    """Host index arrays (numpy / CPU tensors) -> device tensors through ONE non-blocking copy out of a pinned ring slot.
    A `.to(device)` of a pageable host array blocks the host until everything queued on the stream has run (the copy is
    stream-ordered and staged synchronously): four of them per coder launch made the decoder's level loop wait for the
    previous level's launch before it could even build the next one (3.4 ms per level at 1 M anchors)."""
    # (reshape(-1): a 0-d array has no uint8 view — ADVICE r4 — and a scalar buffer of a checkpoint must travel too)
    arrs = [np.ascontiguousarray(a.numpy() if isinstance(a, torch.Tensor) else a) for a in arrays]
    shapes = [a.shape for a in arrs]
    arrs = [a.reshape(-1) for a in arrs]
    sizes = [(a.nbytes + 15) // 16 * 16 for a in arrs]
    total = max(sum(sizes), 16)
    ring = _SMALL_RING
    if key is not None:       # a purpose of its own (e.g. the MLP checkpoint, ~0.5 MB): one grow-only buffer, not a ring slot
        slot = _KEYED_SLOTS.setdefault(key, {"buf": torch.empty(0, dtype=torch.uint8), "ev": None})
        if slot["ev"] is not None:
            slot["ev"].synchronize()
        if slot["buf"].numel() < total:
            slot["buf"] = torch.empty(int(total * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
    elif len(ring["slots"]) < 16:
        ring["slots"].append({"buf": torch.empty(max(total, 1 << 16), dtype=torch.uint8, pin_memory=True), "ev": None})
        slot = ring["slots"][-1]
    else:
        slot = ring["slots"][ring["next"] % 16]
        ring["next"] += 1
        if slot["ev"] is not None:
            slot["ev"].synchronize()              # the copy that last used this slot (16 uploads ago) has long finished
        if slot["buf"].numel() < total:
            slot["buf"] = torch.empty(int(total * 1.5), dtype=torch.uint8, pin_memory=True)
    host = slot["buf"].numpy()
    pos, spans = 0, []
    for a, n, shp in zip(arrs, sizes, shapes):
        host[pos:pos + a.nbytes] = a.view(np.uint8).reshape(-1) if a.nbytes else host[pos:pos]
        spans.append((pos, a.nbytes, a.dtype, shp))
        pos += n
    d = torch.empty(total, dtype=torch.uint8, device=dev)
    d.copy_(slot["buf"][:total], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    slot["ev"] = ev
    out = []
    for (p0, nb, dt, shape) in spans:
        out.append(d[p0:p0 + nb].view(torch.from_numpy(np.empty(0, dtype=dt)).dtype).view(shape))
    return out


def gaussian_encode_packed(x, mean, scale, Q, stream_off, q_div=1, staging=False, lanes=False, overlap=None, deferred=False,
                           stage_key="bitstream", copy_stream=None, ready_out=None):
    """x/mean/scale flat [n] device tensors, element i uses Q[i // q_div]; stream_off int64 [S+1]
    (device or host).  Every stream is coded by its own wave of ONE launch.  Returns (blob, lens, min, max):
    blob = uint8 ndarray holding the S streams back to back (exactly the bytes of the reference's
    b"".join(chunk strings) file), lens int64[S], min/max int32[S] (host).
    lanes (container version 2): every [stream_off[s], stream_off[s+1]) is a BLOCK coded as 64 interleaved lane streams
    behind a 128-byte header of their lengths (csrc/codec.hip, "Lane-parallel Gaussian codec")."""
    L = _lib.lib()
    _lib.require_device(x, mean, scale, Q)
    x, mean, scale, Q = _f(x).reshape(-1), _f(mean).reshape(-1), _f(scale).reshape(-1), _f(Q).reshape(-1)
    dev = x.device
    off = torch.as_tensor(stream_off, dtype=torch.int64)
    S = int(off.numel()) - 1
    if S <= 0:
        return np.zeros(0, np.uint8), np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32)
    off_h = off.cpu()
    off_d = off_h.to(dev)
    stream = _lib.current_stream()
    mnmx = torch.empty(2, S, dtype=torch.int32, device=dev)
    mn, mx = mnmx[0], mnmx[1]
    _lib.check(L.cgs_gaussian_stream_minmax(_lib.ptr(x), _lib.ptr(Q), q_div, _lib.ptr(off_d), S, _lib.ptr(mn),
                                            _lib.ptr(mx), stream), "cgs_gaussian_stream_minmax")
    lens_sym = (off_h[1:] - off_h[:-1])
    if lanes:
        caps = 128 + 64 * (((lens_sym + 63) // 64 * 2 + 16 + 7) // 8 * 8)        # cgs_lanes_block_slot_bytes
    else:
        caps = (lens_sym * 2 + 16 + 7) // 8 * 8
    out_off_h = torch.zeros(S + 1, dtype=torch.int64)
    out_off_h[1:] = torch.cumsum(caps, 0)
    out = torch.empty(int(out_off_h[-1]) + 16, dtype=torch.uint8, device=dev)
    out_off = out_off_h.to(dev)
    # [status, len_0 .. len_{S-1}] in one buffer: ONE device->host read decides everything below
    st_len = torch.zeros(S + 1, dtype=torch.int32, device=dev)
    status, out_len = st_len[:1], st_len[1:]
    enc = L.cgs_gaussian_ac_encode_lanes if lanes else L.cgs_gaussian_ac_encode
    _lib.check(enc(_lib.ptr(x), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(Q), q_div,
                   _lib.ptr(off_d), S, _lib.ptr(mn), _lib.ptr(mx), _lib.ptr(out), _lib.ptr(out_off),
                   _lib.ptr(out_len), _lib.ptr(status), stream), "cgs_gaussian_ac_encode")
    dst_off = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    torch.cumsum(out_len, 0, out=dst_off[1:])
    if overlap is not None:
        overlap()                 # host work of the caller's while the coder launch runs (the read below waits for it)
    st_len_h = st_len.cpu().numpy()
    st = int(st_len_h[0])
    if st != 0:
        raise RuntimeError("gaussian codec: " + ("symbol outside [min,max] / grid wider than 2^16" if st == 1
                                                 else "stream overflowed its buffer"))
    lens = st_len_h[1:].astype(np.int64)
    packed = torch.empty(max(int(lens.sum()), 1), dtype=torch.uint8, device=dev)
    if lanes:
        _lib.check(L.cgs_lanes_compact(_lib.ptr(out), _lib.ptr(out_off), _lib.ptr(off_d), _lib.ptr(dst_off), S,
                                       _lib.ptr(packed), 2, stream), "cgs_lanes_compact")
    else:
        _lib.check(L.cgs_streams_compact(_lib.ptr(out), _lib.ptr(out_off), _lib.ptr(out_len), _lib.ptr(dst_off), S,
                                         _lib.ptr(packed), stream), "cgs_streams_compact")
    nbytes = int(lens.sum())
    global _STAGE_READY
    _STAGE_READY = None
    if staging and nbytes > 0:
        # pinned, reused staging buffer: ~25 GB/s instead of a pageable copy (~5 GB/s, 25 ms for a 1 M-anchor model).
        # The returned blob ALIASES it: valid until the next staging call (the container driver writes the files first).
        # deferred: the download is queued in pieces with an event each and NOT waited for: write_file(..., ready=
        # stage_ready()) lets the writer threads start on a piece the moment it has landed, so the ~5 ms of PCIe and the
        # ~5 ms of page-cache writes of a 1 M-anchor container overlap instead of adding up.
        mnmx_h = mnmx.cpu().numpy()                   # (before the big copies: this read drains the stream)
        stage = _pinned_staging(nbytes, stage_key)
        events = []
        # copy_stream (a side stream): the download runs beside whatever the caller queues on the current stream next — the coder
        # launch of the next range of groups (gaussian_encode_groups(on_range=))
        cs = copy_stream if copy_stream is not None else torch.cuda.current_stream(dev)
        if copy_stream is not None:
            copy_stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cs):
            for lo in range(0, nbytes, _STAGE_PIECE):
                hi = min(nbytes, lo + _STAGE_PIECE)
                stage[lo:hi].copy_(packed[lo:hi], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(cs)
                events.append(ev)
        if copy_stream is not None:
            packed.record_stream(copy_stream)
        ready = StageReady(stage.data_ptr(), nbytes, events, packed)
        if ready_out is not None:
            ready_out.append(ready)
        else:
            _STAGE_READY = ready
        if not deferred:
            ready.wait_all()
            if ready_out is None:
                _STAGE_READY = None
        blob = stage.numpy()[:nbytes]
    else:
        blob = packed.cpu().numpy()[:nbytes]
        mnmx_h = mnmx.cpu().numpy()
    return blob, lens, mnmx_h[0], mnmx_h[1]


def gaussian_encode_streams(x, mean, scale, Q, stream_off, q_div=1):
    """gaussian_encode_packed with the streams as S separate byte strings:
    (list of S byte strings, min int32[S] host, max int32[S] host)."""
    blob, lens, mn, mx = gaussian_encode_packed(x, mean, scale, Q, stream_off, q_div)
    ends = np.cumsum(lens)
    return [blob[e - n: e].tobytes() for e, n in zip(ends, lens)], mn, mx


def _pack_flat(parts, dev):
    """[tensor | (Q rows, q_div)] -> one flat float32 device tensor holding them back to back, each written ONCE into its
    slice (a strided slice of the prediction, or a per-row step size broadcast over its q_div elements, used to be made
    contiguous / expanded first and then copied again by torch.cat: ~1 GB of extra traffic per coder launch at 1 M anchors)."""
    sizes = [int(p[0].numel()) * int(p[1]) if isinstance(p, tuple) else int(p.numel()) for p in parts]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    base = 0
    for p, n in zip(parts, sizes):
        if n:
            if isinstance(p, tuple):
                q, q_div = p[0].reshape(-1), int(p[1])
                flat[base:base + n].view(q.numel(), q_div).copy_(q.unsqueeze(1).expand(q.numel(), q_div))
            else:
                flat[base:base + n].view(p.shape).copy_(p)
        base += n
    return flat, sizes


def _expand_q(Q, q_div):
    Q = _f(Q).reshape(-1)
    return Q if q_div == 1 else Q.repeat_interleave(int(q_div))


RANGE_MIN_SYMBOLS = 16 << 20          # below this many symbols the groups go through one coder launch


def _group_ranges(groups):
    """Consecutive runs of groups, each at most ~40 % of the symbols (a group larger than that is a run of its own)."""
    sizes = [int(g_[0].numel()) for g_ in groups]
    total = sum(sizes)
    runs, cur, acc = [], [], 0
    for i, n in enumerate(sizes):
        if cur and acc + n > 0.4 * total:
            runs.append(cur); cur, acc = [], 0
        cur.append(i); acc += n
    if cur:
        runs.append(cur)
    return runs, total


_COPY_STREAMS = {}


def gaussian_encode_groups(groups, staging=False, lanes=False, overlap=None, deferred=False, on_range=None):
    """groups = [(x, mean, scale, Q, stream_off, q_div), ...] -> [(blob, lens, min, max), ...] (see gaussian_encode_packed).
    staging: download through the module's reused pinned buffer; the blobs then alias it until the next staging call.
    All streams of all groups go through ONE coder launch: a stream is a serial chain on one wave, so the launch
    lasts as long as its longest stream however many streams it holds.
    on_range (with staging + deferred, one process, a large model): the groups are coded in 2-4 consecutive RANGES, a launch
    each, and on_range(indices, results, ready) is called as soon as a range's stream lengths are on the host — its download
    (a side stream, a pinned buffer of its own) and whatever the callback starts (the file writers) then run beside the coder
    launch of the next range instead of behind the last one.  Same bytes: a range's streams are the same streams."""
    groups = list(groups)
    if not groups:
        return []
    from . import dist as _D0
    if on_range is not None and staging and deferred and _D0.world() == 1:
        runs, total = _group_ranges(groups)
        if len(runs) > 1 and total >= RANGE_MIN_SYMBOLS:
            dev = groups[0][0].device
            cs = _COPY_STREAMS.get(dev)
            if cs is None:
                cs = _COPY_STREAMS[dev] = torch.cuda.Stream(device=dev)
            out = [None] * len(groups)
            for r, idxs in enumerate(runs):
                sub = [groups[i] for i in idxs]
                ready_out = []
                res = _encode_group_run(sub, lanes, overlap if r == 0 else None, f"bitstream{r}", cs, ready_out)
                for i, one in zip(idxs, res):
                    out[i] = one
                on_range(idxs, res, ready_out[0] if ready_out else None)
            return out
    return _encode_group_run(groups, lanes, overlap, "bitstream", None, None, staging=staging, deferred=deferred)


def _encode_group_run(groups, lanes, overlap, stage_key, copy_stream, ready_out, staging=True, deferred=True):
    """One coder launch over `groups` (see gaussian_encode_groups)."""
    xs, ms, ss, qs, edges, counts, base = [], [], [], [], [torch.zeros(1, dtype=torch.int64)], [], 0
    dev = groups[0][0].device
    _lib.require_device(*[t for g_ in groups for t in g_[:4]])
    for (x, mean, scale, Q, off, q_div) in groups:
        off = torch.as_tensor(off, dtype=torch.int64).cpu()
        xs.append(x); ms.append(mean); ss.append(scale)
        qs.append((Q, int(q_div)))
        n_sym = int(off[-1]) if off.numel() else 0
        if not (x.numel() == mean.numel() == scale.numel() == Q.numel() * int(q_div) == n_sym) or (off.numel() and int(off[0])):
            raise ValueError("gaussian_encode_groups: a group's x / mean / scale / Q sizes do not match its stream offsets")
        edges.append(off[1:] + base)
        counts.append(max(int(off.numel()) - 1, 0))
        base += int(x.numel())
    (X, _), (M, _), (Sc, _), (Qe, _) = (_pack_flat(p_, dev) for p_ in (xs, ms, ss, qs))
    E = torch.cat(edges)
    from . import dist as D
    if D.world() > 1:
        # multi-GPU: rank r codes a contiguous block of the stream list; the ranks' packed bytes in rank
        # order ARE the single-GPU bytes, so rank 0 just concatenates.  No exchange is needed while encoding (every
        # rank predicts and quantises all levels itself; the coder never feeds back into the context).
        b = D.stream_blocks(E)
        s0, s1 = b[D.rank()], b[D.rank() + 1]
        e0, e1 = int(E[s0]), int(E[s1])
        part = gaussian_encode_packed(X[e0:e1], M[e0:e1], Sc[e0:e1], Qe[e0:e1], E[s0:s1 + 1] - e0, 1, lanes=lanes)
        parts = D.gather_objects(part, dst=0)
        if parts is None:
            return None
        blob, lens, mn, mx = (np.concatenate([p[i] for p in parts]) for i in range(4))
    else:
        blob, lens, mn, mx = gaussian_encode_packed(X, M, Sc, Qe, E, 1, staging=staging, lanes=lanes, overlap=overlap, deferred=deferred,
                                                    stage_key=stage_key, copy_stream=copy_stream, ready_out=ready_out)
        overlap = None
    if overlap is not None:
        overlap()
    out, s0, b0 = [], 0, 0
    for c in counts:
        nb = int(lens[s0:s0 + c].sum())
        out.append((blob[b0:b0 + nb], lens[s0:s0 + c], mn[s0:s0 + c], mx[s0:s0 + c]))
        s0 += c; b0 += nb
    return out


def gaussian_decode_packed(mean, scale, Q, stream_off, min_v, max_v, blob, lens, q_div=1, lanes=False):
    """Inverse of gaussian_encode_packed -> flat float32 [n] device tensor of dequantised values.
    blob: bytes / uint8 ndarray with the S streams back to back; lens: their byte lengths."""
    L = _lib.lib()
    _lib.require_device(mean, scale, Q)
    mean, scale, Q = _f(mean).reshape(-1), _f(scale).reshape(-1), _f(Q).reshape(-1)
    dev = mean.device
    off_h = torch.as_tensor(stream_off, dtype=torch.int64).cpu()
    S = int(off_h.numel()) - 1
    x_out = torch.empty(mean.numel(), dtype=torch.float32, device=dev)
    if S <= 0:
        return x_out
    lens = np.asarray(lens, dtype=np.int64)
    assert lens.shape[0] == S
    in_off_h = np.zeros(S + 1, dtype=np.int64)
    np.cumsum(lens, out=in_off_h[1:])
    if isinstance(blob, torch.Tensor) and blob.is_cuda:
        # already on the device (StagedFiles): a uint8 slice followed by >= 16 readable bytes of its storage
        assert blob.dtype == torch.uint8 and blob.numel() == int(in_off_h[-1]), "stream lengths do not add up to the blob"
        in_d = blob
    else:
        buf = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
        assert buf.size == int(in_off_h[-1]), "stream lengths do not add up to the blob"
        in_d = torch.empty(buf.size + 16, dtype=torch.uint8, device=dev)
        if buf.size:
            in_d[: buf.size].copy_(torch.from_numpy(np.ascontiguousarray(buf) if buf.flags.writeable else buf.copy()))
    in_off, mn, mx, off_d = _upload_small([in_off_h, np.asarray(min_v, dtype=np.int32), np.asarray(max_v, dtype=np.int32), off_h],
                                          dev)
    if lanes:
        _lib.check(L.cgs_gaussian_ac_decode_lanes(_lib.ptr(mean), _lib.ptr(scale), _lib.ptr(Q), q_div, _lib.ptr(off_d), S,
                                                  _lib.ptr(mn), _lib.ptr(mx), _lib.ptr(in_d), _lib.ptr(in_off), _lib.ptr(x_out),
                                                  _lib.ptr(decode_status(dev)), _lib.current_stream()), "cgs_gaussian_ac_decode_lanes")
    else:
        _lib.check(L.cgs_gaussian_ac_decode(_lib.ptr(mean), _lib.ptr(scale), _lib.ptr(Q), q_div, _lib.ptr(off_d), S,
                                            _lib.ptr(mn), _lib.ptr(mx), _lib.ptr(in_d), _lib.ptr(in_off), _lib.ptr(x_out),
                                            _lib.current_stream()), "cgs_gaussian_ac_decode")
    return x_out


_DECODE_STATUS = {}


def decode_status(device, reset=False):
    """The device int32 the lane decoders of container version 2 report a malformed block into (1 + block index of the launch
    that saw it; 0 = fine).  conduct_decoding zeroes it first and reads it once at the end (decode_status_check)."""
    key = str(device)
    st = _DECODE_STATUS.get(key)
    if st is None:
        st = _DECODE_STATUS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    elif reset:
        st.zero_()
    return st


def decode_status_check(device):
    bad = int(decode_status(device).item())
    if bad:
        decode_status(device, reset=True)
        raise ValueError(f"container version 2: block {bad - 1} of a lane-coded file is malformed (its lane lengths do not add up to "
                         "the block length of the header, or its min / max are out of range): truncated or corrupt container")


def gaussian_decode_streams(mean, scale, Q, stream_off, min_v, max_v, streams, q_div=1):
    """gaussian_decode_packed on S separate byte strings."""
    return gaussian_decode_packed(mean, scale, Q, stream_off, min_v, max_v, b"".join(streams),
                                  [len(b) for b in streams], q_div)


def _join_device_slices(parts):
    """cat of uint8 device tensors; free when they are consecutive slices of one storage (StagedFiles hands them out so)."""
    parts = [p_ for p_ in parts if p_.numel() > 0] or parts[:1]
    first = parts[0]
    end = first.data_ptr() + first.numel()
    for p_ in parts[1:]:
        if p_.untyped_storage().data_ptr() != first.untyped_storage().data_ptr() or p_.data_ptr() != end:
            return torch.cat(parts + [torch.zeros(16, dtype=torch.uint8, device=first.device)])[:-16]
        end += p_.numel()
    total = sum(int(p_.numel()) for p_ in parts)
    return torch.empty(0, dtype=torch.uint8, device=first.device).set_(first.untyped_storage(), first.storage_offset(), (total,), (1,))


def file_ranges(write: bool, ranges, threads: int = 8):
    """ranges = [(path, file offset, byte count, host address), ...] read (write=False) or written by C++ workers
    (cgs_pread_ranges / cgs_pwrite_ranges): one call, the GIL released for its duration."""
    ranges = [r for r in ranges if r[2] > 0]
    if not ranges:
        return
    n = len(ranges)
    paths = (C.c_char_p * n)(*[os.fsencode(r[0]) for r in ranges])
    foff = (C.c_int64 * n)(*[int(r[1]) for r in ranges])
    nb = (C.c_int64 * n)(*[int(r[2]) for r in ranges])
    mem = (C.c_void_p * n)(*[int(r[3]) for r in ranges])
    L = _lib.lib()
    fn = L.cgs_pwrite_ranges if write else L.cgs_pread_ranges
    _lib.check(fn(n, paths, foff, nb, mem, int(threads)), "cgs_pwrite_ranges" if write else "cgs_pread_ranges")


_PINNED = {}
_LAST_STAGED = {}
_STAGE_POOL = None


def _wait_staged(released, job, events):
    """Host-side wait for a StagedFiles' reads and copies (see StagedFiles.wait_all)."""
    released.set()
    try:
        job.result()
    except BaseException:
        pass
    for ev in list(events.values()):
        ev.synchronize()


def _stage_pool():
    """A few threads of their own for file staging (the shared pool may be full of hyper-prior rANS jobs)."""
    global _STAGE_POOL
    if _STAGE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _STAGE_POOL = ThreadPoolExecutor(max_workers=2, thread_name_prefix="cgs-stage")
    return _STAGE_POOL


_STAGE_STREAMS = {}


class StagedFiles:
    """The bitstream files of a container, read into ONE pinned host buffer and copied to ONE device buffer by a host
    thread on a side stream while the caller does something else (conduct_decoding: the hyper prior's rANS strings and the
    first levels).  get(name) waits for the copy and returns that file's bytes as a device uint8 slice; consecutive
    names are consecutive slices, so the streams of one coder launch need no concatenation.  Replaces, per file,
    np.fromfile + np.concatenate + a pageable host-to-device copy (three passes over ~120 MB at 1 M anchors)."""

    def __init__(self, paths, device, start=None):
        """start: number of leading files whose reads begin at once; the others wait for release() (the decoder reads its
        checkpoint in between, on a quiet interpreter)."""
        self.names = list(paths)
        sizes = [os.path.getsize(p_) for p_ in self.names]
        self.off = {}
        pos = 0
        for p_, n in zip(self.names, sizes):
            self.off[p_] = (pos, n)
            pos += n
        self.total = pos
        self.device = device
        # ONE staging stream per device for the life of the process: the caching allocator keeps a pool per stream, so a fresh
        # stream per container meant a fresh ~120 MB hipMalloc for dev_buf in front of every decode (~5 ms on its critical path)
        skey = str(device)
        self.stream = _STAGE_STREAMS.get(skey)
        if self.stream is None:
            self.stream = _STAGE_STREAMS[skey] = torch.cuda.Stream(device=device)
        # allocated UNDER the side stream (the caching allocator then never hands out a block that kernels queued on
        # the caller's stream may still be reading); consumers on other streams are registered in get()
        self.stream.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(self.stream):
            self.dev_buf = torch.empty(pos + 64, dtype=torch.uint8, device=device)
        import threading
        self.ready = {p_: threading.Event() for p_ in self.names}     # host side: the file's copies have been enqueued
        self.events = {}                                              # device side: the file's bytes have arrived
        self.error = None
        key = (self.device.index if isinstance(self.device, torch.device) else 0)
        # the previous container's copies read the pinned buffer this one is about to overwrite: wait for them.  Only what that
        # takes is kept from it — its release switch, its job and its events (keeping the object itself pinned
        # ~120 MB of device memory, the whole container, for the life of the process)
        prev = _LAST_STAGED.pop(key, None)
        if prev is not None:
            _wait_staged(*prev)
        pinned = _PINNED.get(key)
        if pinned is None or pinned.numel() < self.total + 64:
            pinned = _PINNED[key] = torch.empty(max(int(self.total * 1.25) + 64, 1 << 20), dtype=torch.uint8, pin_memory=True)
        self._pinned = pinned
        # ONE interpreter thread drives the staging; the reads themselves are plain C++ workers inside cgs_pread_ranges
        # (csrc/file_io.cpp: 8 MB ranges, eight workers — one thread copies ~8 GB/s out of the page cache and the 120 MB of
        # a 1 M-anchor container were the decoder's longest chain).  Eight interpreter threads calling os.preadv did the
        # same reads, but each finished piece made its thread queue for the GIL and the decoder's prologue on the main
        # thread ran 3-10 x slower beside them.  Files are staged in the order the coder launches consume them, in batches
        # of <= 32 MB, each followed by its host-to-device copy; a file's event is recorded once its last byte is queued.
        self._released = threading.Event()
        self._start = len(self.names) if start is None else max(0, int(start))
        if self._start >= len(self.names):
            self._released.set()
        self._job = _stage_pool().submit(self._run)
        _LAST_STAGED[key] = (self._released, self._job, self.events)

    def release(self):
        """Start the reads held back by `start`."""
        self._released.set()

    def _run(self):
        try:
            piece, batch_bytes = 8 << 20, 32 << 20
            base = self._pinned.data_ptr()
            i = 0
            while i < len(self.names):
                if i >= self._start:
                    # (bounded: a caller that fails between construction and release() must not leave this pool thread — which
                    #  the interpreter joins at exit — waiting for ever; after the timeout the files are simply staged)
                    self._released.wait(10.0)
                    self._released.set()
                # a batch: whole files from i on, up to batch_bytes (one file alone may exceed it: it is then split)
                stop = len(self.names) if self._released.is_set() else self._start
                j, size = i, 0
                while j < stop and (j == i or size + self.off[self.names[j]][1] <= batch_bytes):
                    size += self.off[self.names[j]][1]
                    j += 1
                lo0 = self.off[self.names[i]][0]
                for b0 in range(0, max(size, 1), batch_bytes):
                    b1 = min(size, b0 + batch_bytes)
                    ranges = []
                    for p_ in self.names[i:j]:
                        pos, n = self.off[p_]
                        f0, f1 = max(pos, lo0 + b0), min(pos + n, lo0 + b1)      # this file's part of [b0, b1)
                        for q in range(f0, f1, piece):
                            ranges.append((p_, q - pos, min(f1, q + piece) - q, base + q))
                    file_ranges(False, ranges, 8)
                    if b1 > b0:
                        with torch.cuda.stream(self.stream):
                            self.dev_buf[lo0 + b0:lo0 + b1].copy_(self._pinned[lo0 + b0:lo0 + b1], non_blocking=True)
                with torch.cuda.stream(self.stream):
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                for p_ in self.names[i:j]:
                    self.events[p_] = ev
                    self.ready[p_].set()
                i = j
        except BaseException as e:    # wake the waiters, get() re-raises
            self.error = e
            for r in self.ready.values():
                r.set()
            raise
        return True

    def wait_all(self):
        """Host-side: every copy has finished (the pinned buffer is reused by the next container)."""
        _wait_staged(self._released, self._job, self.events)

    def get(self, name):
        if not self.ready[name].is_set() and self.names.index(name) >= self._start:
            self.release()
        self.ready[name].wait()
        if self.error is not None:
            raise self.error
        cur = torch.cuda.current_stream()
        cur.wait_event(self.events[name])
        self.dev_buf.record_stream(cur)       # the buffer belongs to the side stream's pool: keep it alive for this consumer
        pos, n = self.off[name]
        return self.dev_buf[pos:pos + n]


def gaussian_decode_groups(groups, lanes=False):
    """groups = [(mean, scale, Q, stream_off, min_v, max_v, blob, lens, q_div), ...] -> [flat float32 values, ...];
    one coder launch for all streams of all groups."""
    if not groups:
        return []
    ms, ss, qs, edges, mns, mxs, blobs, lns, base = [], [], [], [torch.zeros(1, dtype=torch.int64)], [], [], [], [], 0
    dev = groups[0][0].device
    _lib.require_device(*[t for g_ in groups for t in g_[:3]])
    for (mean, scale, Q, off, mn, mx, blob, lens, q_div) in groups:
        off = torch.as_tensor(off, dtype=torch.int64).cpu()
        if not (mean.numel() == scale.numel() == Q.numel() * int(q_div)):
            raise ValueError("gaussian_decode_groups: a group's mean / scale / Q sizes do not match")
        ms.append(mean); ss.append(scale); qs.append((Q, int(q_div)))
        edges.append(off[1:] + base)
        mns.append(np.asarray(mn, dtype=np.int32).reshape(-1)); mxs.append(np.asarray(mx, dtype=np.int32).reshape(-1))
        blobs.append(blob if isinstance(blob, (np.ndarray, torch.Tensor)) else np.frombuffer(blob, dtype=np.uint8))
        lns.append(np.asarray(lens, dtype=np.int64).reshape(-1))
        base += int(mean.numel())
    (M, sizes), (Sc, _), (Qe, _) = (_pack_flat(p_, dev) for p_ in (ms, ss, qs))
    E = torch.cat(edges)
    mn, mx, lens = np.concatenate(mns), np.concatenate(mxs), np.concatenate(lns)
    if all(isinstance(b_, torch.Tensor) for b_ in blobs):
        blob = _join_device_slices(blobs)
    else:
        blob = np.concatenate([b_.cpu().numpy() if isinstance(b_, torch.Tensor) else b_ for b_ in blobs])
    from . import dist as D
    if D.world() > 1:
        # multi-GPU: rank r decodes a contiguous block of the streams, then ONE all-gather hands every rank all
        # decoded values (they are the next level's context and the final parameters on every replica)
        b = D.stream_blocks(E)
        s0, s1 = b[D.rank()], b[D.rank() + 1]
        e0, e1 = int(E[s0]), int(E[s1])
        cum = np.concatenate([[0], np.cumsum(lens)])
        local = gaussian_decode_packed(M[e0:e1], Sc[e0:e1], Qe[e0:e1], E[s0:s1 + 1] - e0, mn[s0:s1], mx[s0:s1],
                                       blob[int(cum[s0]):int(cum[s1])], lens[s0:s1], 1, lanes=lanes)
        flat = D.all_gather_rows(local, [int(E[b[r + 1]]) - int(E[b[r]]) for r in range(D.world())])
    else:
        flat = gaussian_decode_packed(M, Sc, Qe, E, mn, mx, blob, lens, 1, lanes=lanes)
    return list(torch.split(flat, sizes))


# ---- lane-parallel table codec (container version 2: hyper.b) ------------------------------------------------------
def _table_blocks(C_, N, block):
    """Blocks of <= `block` consecutive anchors of one channel, channel-major: (edges int64 [S+1], channel int32 [S])."""
    per = list(range(0, N, block)) + [N] if N > 0 else [0]
    edges, ch = [0], []
    for c in range(C_):
        for a, b in zip(per[:-1], per[1:]):
            edges.append(c * N + b)
            ch.append(c)
    return torch.tensor(edges, dtype=torch.int64), torch.tensor(ch, dtype=torch.int32)


def table_encode_lanes(sym, cdf, cdf_len, offset, block):
    """sym int32 [C, N] device symbols; cdf / cdf_len / offset: EntropyBottleneck's device tables.  -> (blob uint8 ndarray of the
    blocks back to back, lens int64 [S]); one launch, one wave per block, one coder per lane."""
    L = _lib.lib()
    _lib.require_device(sym, cdf, cdf_len, offset)
    sym = sym.to(torch.int32).contiguous()
    C_, N = int(sym.shape[0]), int(sym.shape[1])
    edges_h, ch_h = _table_blocks(C_, N, int(block))
    S = int(ch_h.numel())
    if S == 0:
        return np.zeros(0, np.uint8), np.zeros(0, np.int64)
    dev = sym.device
    lens_sym = edges_h[1:] - edges_h[:-1]
    caps = 128 + 64 * (((lens_sym + 63) // 64 * 6 + 16 + 7) // 8 * 8)                       # cgs_lanes_block_slot_bytes(n, 6)
    out_off_h = torch.zeros(S + 1, dtype=torch.int64)
    out_off_h[1:] = torch.cumsum(caps, 0)
    edges, ch, out_off = edges_h.to(dev), ch_h.to(dev), out_off_h.to(dev)
    out = torch.empty(int(out_off_h[-1]) + 16, dtype=torch.uint8, device=dev)
    st_len = torch.zeros(S + 1, dtype=torch.int32, device=dev)
    cdf = cdf.to(torch.int32).contiguous()
    cl, of = cdf_len.to(torch.int32).contiguous(), offset.to(torch.int32).contiguous()
    stream = _lib.current_stream()
    _lib.check(L.cgs_table_ac_encode_lanes(_lib.ptr(sym), _lib.ptr(edges), _lib.ptr(ch), S, _lib.ptr(cdf), int(cdf.shape[1]),
                                           _lib.ptr(cl), _lib.ptr(of), _lib.ptr(out), _lib.ptr(out_off), _lib.ptr(st_len[1:]),
                                           _lib.ptr(st_len[:1]), stream), "cgs_table_ac_encode_lanes")
    dst_off = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    torch.cumsum(st_len[1:], 0, out=dst_off[1:])
    st_len_h = st_len.cpu().numpy()
    if int(st_len_h[0]) != 0:
        raise RuntimeError("table codec: a lane slot overflowed (latents far outside the prior's support)")
    lens = st_len_h[1:].astype(np.int64)
    packed = torch.empty(max(int(lens.sum()), 1), dtype=torch.uint8, device=dev)
    _lib.check(L.cgs_lanes_compact(_lib.ptr(out), _lib.ptr(out_off), _lib.ptr(edges), _lib.ptr(dst_off), S, _lib.ptr(packed), 6,
                                   stream), "cgs_lanes_compact")
    return packed.cpu().numpy()[: int(lens.sum())], lens


def table_decode_lanes(blob, lens, C_, N, cdf, cdf_len, offset, medians, block):
    """Inverse of table_encode_lanes -> dequantised rows [N, C] float32 on the tables' device (symbol + medians[c]).
    blob: bytes / uint8 ndarray / uint8 device tensor followed by >= 16 readable bytes."""
    L = _lib.lib()
    dev = cdf.device
    out = torch.empty(N, C_, dtype=torch.float32, device=dev)
    edges_h, ch_h = _table_blocks(C_, N, int(block))
    S = int(ch_h.numel())
    if S == 0:
        return out
    lens = np.asarray(lens, dtype=np.int64)
    assert lens.shape[0] == S, "hyper.b: block count does not match the header"
    if isinstance(blob, torch.Tensor) and blob.is_cuda:
        in_d = blob
    else:
        buf = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
        in_d = torch.zeros(buf.size + 16, dtype=torch.uint8, device=dev)
        if buf.size:
            in_d[: buf.size].copy_(torch.from_numpy(np.ascontiguousarray(buf) if buf.flags.writeable else buf.copy()))
    in_off_h = np.zeros(S + 1, dtype=np.int64)
    np.cumsum(lens, out=in_off_h[1:])
    assert int(in_off_h[-1]) <= int(in_d.numel())
    # (every operand is held in a local until the launch is enqueued: a temporary's block goes back to the caching allocator the
    #  moment its pointer has been read, and the next temporary's upload would land in it ahead of the kernel)
    cdf = cdf.to(torch.int32).contiguous()
    edges, ch, in_off = _upload_small([edges_h, ch_h, in_off_h], dev)
    cl, of, med = cdf_len.to(torch.int32).contiguous(), offset.to(torch.int32).contiguous(), medians.to(torch.float32).contiguous()
    _lib.check(L.cgs_table_ac_decode_lanes(_lib.ptr(edges), _lib.ptr(ch), S, _lib.ptr(cdf), int(cdf.shape[1]), _lib.ptr(cl),
                                           _lib.ptr(of), _lib.ptr(med), N, _lib.ptr(in_d), _lib.ptr(in_off), _lib.ptr(out), C_,
                                           _lib.ptr(decode_status(out.device)), _lib.current_stream()), "cgs_table_ac_decode_lanes")
    return out


def gaussian_cdf_table(mean, scale, Q, min_v, max_v, q_div=1):
    """uint16 [n, max-min+2] integer CDF table of one stream (test hook)."""
    L = _lib.lib()
    mean, scale, Q = _f(mean).reshape(-1), _f(scale).reshape(-1), _f(Q).reshape(-1)
    n, Lp = mean.numel(), int(max_v) - int(min_v) + 2
    table = torch.empty(n, Lp, dtype=torch.int16, device=mean.device)
    _lib.check(L.cgs_gaussian_cdf_table(_lib.ptr(mean), _lib.ptr(scale), _lib.ptr(Q), q_div, n, int(min_v), int(max_v),
                                        _lib.ptr(table), _lib.current_stream()), "cgs_gaussian_cdf_table")
    return table.cpu().numpy().view(np.uint16)


# ---- the reference's single-stream signatures (utils/encodings.py:83-144) -----------------------------
def encoder_gaussian(x, mean, scale, Q, file_name=None):
    if file_name is not None:
        assert file_name.endswith(".b")
    if not isinstance(Q, torch.Tensor):
        Q = torch.tensor([Q], dtype=mean.dtype, device=mean.device).repeat(mean.shape[0])
    assert x.shape == mean.shape == scale.shape == Q.shape
    n = x.numel()
    streams, mn, mx = gaussian_encode_streams(x, mean, scale, Q, [0, n])
    byte_stream = streams[0]
    if file_name is not None:
        with open(file_name, "wb") as f:
            f.write(byte_stream)
    dev = x.device
    return (byte_stream, len(byte_stream) * 8, torch.tensor(float(mn[0]), device=dev), torch.tensor(float(mx[0]), device=dev))


def decoder_gaussian(mean, scale, Q, file_name=None, min_value=-100, max_value=100, bstream=None):
    if file_name is not None:
        assert file_name.endswith(".b")
        with open(file_name, "rb") as f:
            bstream = f.read()
    else:
        assert bstream is not None
    if not isinstance(Q, torch.Tensor):
        Q = torch.tensor([Q], dtype=mean.dtype, device=mean.device).repeat(mean.shape[0])
    assert mean.shape == scale.shape == Q.shape
    mn = int(min_value.item()) if isinstance(min_value, torch.Tensor) else int(min_value)
    mx = int(max_value.item()) if isinstance(max_value, torch.Tensor) else int(max_value)
    return gaussian_decode_streams(mean, scale, Q, [0, mean.numel()], [mn], [mx], [bstream]).reshape(mean.shape)


# ---- hyper-prior rANS ------------------------------------------------------------------------------------
def rans_encode_channels(symbols: np.ndarray, cdf: np.ndarray, cdf_len: np.ndarray, offset: np.ndarray,
                         precision: int = 16) -> bytes:
    """symbols int32 [C, n]; cdf int32 [C, max_len]; returns one byte string."""
    L = _lib.lib()
    sym = np.ascontiguousarray(symbols, dtype=np.int32)
    cdf = np.ascontiguousarray(cdf, dtype=np.int32)
    cl = np.ascontiguousarray(cdf_len, dtype=np.int32)
    of = np.ascontiguousarray(offset, dtype=np.int32)
    Cn, n = sym.shape
    cap = L.cgs_rans_max_bytes(Cn * n)
    out = np.empty(cap, dtype=np.uint8)
    ln = C.c_size_t(0)
    _lib.check(L.cgs_rans_encode_host(_np_ptr(sym), Cn, n, _np_ptr(cdf), cdf.shape[1], _np_ptr(cl), _np_ptr(of),
                                      precision, _np_ptr(out), cap, C.byref(ln)), "cgs_rans_encode_host")
    return out[: ln.value].tobytes()


def rans_decode_channels(data: bytes, channels: int, n: int, cdf: np.ndarray, cdf_len: np.ndarray, offset: np.ndarray,
                         precision: int = 16) -> np.ndarray:
    L = _lib.lib()
    buf = np.frombuffer(data, dtype=np.uint8)
    cdf = np.ascontiguousarray(cdf, dtype=np.int32)
    cl = np.ascontiguousarray(cdf_len, dtype=np.int32)
    of = np.ascontiguousarray(offset, dtype=np.int32)
    out = np.empty((channels, n), dtype=np.int32)
    _lib.check(L.cgs_rans_decode_host(_np_ptr(buf), buf.size, channels, n, _np_ptr(cdf), cdf.shape[1], _np_ptr(cl),
                                      _np_ptr(of), precision, _np_ptr(out)), "cgs_rans_decode_host")
    return out
