"""cgs_pread_ranges / cgs_pwrite_ranges (csrc/file_io.cpp: the container's file reads and writes on C++ workers) through
codec.file_ranges and codec.write_file.  Host only — no device call."""
import os

import numpy as np
import pytest

from contextgs_amd import codec


def test_ranges_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, 5_000_003, dtype=np.uint8)
    b = rng.integers(0, 256, 70_001, dtype=np.uint8)
    pa, pb = str(tmp_path / "a.b"), str(tmp_path / "b.b")
    piece = 1 << 20
    ranges = [(pa, lo, min(a.size, lo + piece) - lo, a.ctypes.data + lo) for lo in range(0, a.size, piece)]
    ranges += [(pb, 0, b.size, b.ctypes.data), (pb, 0, 0, 0)]                      # an empty range is skipped
    codec.file_ranges(True, ranges, threads=5)
    assert os.path.getsize(pa) == a.size and os.path.getsize(pb) == b.size
    assert np.array_equal(np.fromfile(pa, dtype=np.uint8), a) and np.array_equal(np.fromfile(pb, dtype=np.uint8), b)
    out = np.zeros(a.size + b.size, dtype=np.uint8)
    rd = [(pa, lo, min(a.size, lo + piece) - lo, out.ctypes.data + lo) for lo in range(0, a.size, piece)]
    rd.append((pb, 0, b.size, out.ctypes.data + a.size))
    codec.file_ranges(False, rd, threads=3)
    assert np.array_equal(out[:a.size], a) and np.array_equal(out[a.size:], b)
    # a slice in the middle of a file
    mid = np.zeros(1000, dtype=np.uint8)
    codec.file_ranges(False, [(pa, 123_456, 1000, mid.ctypes.data)], threads=1)
    assert np.array_equal(mid, a[123_456:124_456])


def test_errors_are_reported(tmp_path):
    buf = np.zeros(16, dtype=np.uint8)
    with pytest.raises(RuntimeError, match="No such file"):
        codec.file_ranges(False, [(str(tmp_path / "missing.b"), 0, 16, buf.ctypes.data)])
    p = str(tmp_path / "short.b")
    np.arange(8, dtype=np.uint8).tofile(p)
    with pytest.raises(RuntimeError, match="shorter"):
        codec.file_ranges(False, [(p, 0, 16, buf.ctypes.data)])


def test_write_file_over_an_older_longer_file(tmp_path):
    p = str(tmp_path / "feat0.b")
    np.zeros(40 << 20, dtype=np.uint8).tofile(p)                                     # the previous container's file
    blob = np.random.default_rng(2).integers(0, 256, (17 << 20) + 5, dtype=np.uint8)
    for job in codec.write_file(p, blob):
        job.result()
    assert os.path.getsize(p) == blob.size and np.array_equal(np.fromfile(p, dtype=np.uint8), blob)
    small = blob[:1000]
    for job in codec.write_file(p, small):
        job.result()
    assert np.array_equal(np.fromfile(p, dtype=np.uint8), small)
