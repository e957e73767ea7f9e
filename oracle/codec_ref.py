"""Pure-Python ORACLE of the entropy coders (small cases only).

TEST INFRASTRUCTURE ONLY — never imported by contextgs_amd/.

PARITY UNPINNED: the reference's coders are the pip wheels `torchac` and `compressai`,
neither in the mount nor pinned (environment.yml:21-22).  This file restates
the PUBLISHED algorithms:

  * float CDF -> 16-bit integer CDF: round(c * (2^16 - (Lp-1))) + j   (call sites
    utils/encodings.py:108,138 pass Lp = max-min+2 columns);
  * a 32-bit low/high binary arithmetic coder with underflow counting, MSB-first bit
    packing, the last symbol's upper bound fixed at 2^16, one trailing bit + padding;
  * the Gaussian table of utils/encodings.py:88-97;
  * a rANS coder with escape + Elias-gamma bypass for the hyper prior (our own format).

It is pinned by known-answer streams worked out by hand in tests/test_codec.py and by
round trips; the product coder must match it bit for bit on identical integer CDFs.
Written with Python ints and explicit bit lists so that it shares no structure with the
C++/HIP implementation.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import erf

HALF, QUARTER, THREEQ, FULL = 1 << 31, 1 << 30, 3 << 30, (1 << 32) - 1


def float_cdf_to_int(cdf_row):
    """One row of floats in [0,1] -> list of uint16 ints."""
    Lp = len(cdf_row)
    scale = np.float32(65536 - (Lp - 1))
    out = []
    for j, c in enumerate(cdf_row):
        v = int(np.rint(np.float32(c) * scale))          # half-to-even, fp32 product
        out.append((v + j) & 0xFFFF)
    return out


def gaussian_table(mean, scale, Q, min_v, max_v):
    """utils/encodings.py:88-97: lower[i, j] = Normal(mean_i, scale_i).cdf((min+j-0.5) * Q_i), fp32."""
    f32 = np.float32
    mean, scale, Q = (np.asarray(v, f32).reshape(-1, 1) for v in (mean, scale, Q))
    samples = np.arange(min_v, max_v + 2, dtype=f32)[None, :]
    z = (((samples - f32(0.5)) * Q - mean) * (f32(1) / scale) / f32(math.sqrt(2))).astype(f32)
    return (f32(0.5) * (f32(1) + erf(z))).astype(f32)


def ac_encode(int_cdf_rows, symbols):
    """int_cdf_rows[i] = list of Lp ints; symbols[i] in [0, Lp-2].  Returns bytes."""
    low, high, pending = 0, FULL, 0
    bits = []

    def emit(b):
        nonlocal pending
        bits.append(b)
        bits.extend([1 - b] * pending)
        pending = 0

    for row, s in zip(int_cdf_rows, symbols):
        top = len(row) - 2
        c_lo = row[s]
        c_hi = 65536 if s == top else row[s + 1]
        span = high - low + 1
        high = (low - 1 + ((span * c_hi) >> 16)) & FULL
        low = (low + ((span * c_lo) >> 16)) & FULL
        while True:
            if high < HALF:
                emit(0)
            elif low >= HALF:
                emit(1)
            elif low >= QUARTER and high < THREEQ:
                pending += 1
                low -= QUARTER
                high -= QUARTER
            else:
                break
            low = (low << 1) & FULL
            high = ((high << 1) | 1) & FULL
    pending += 1
    emit(0 if low < QUARTER else 1)
    while len(bits) % 8:
        bits.append(0)
    return bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))


def ac_decode(int_cdf_rows, data):
    bits = [(byte >> (7 - k)) & 1 for byte in data for k in range(8)]
    pos = 0

    def nxt():
        nonlocal pos
        b = bits[pos] if pos < len(bits) else 0
        pos += 1
        return b

    low, high, value = 0, FULL, 0
    for _ in range(32):
        value = (value << 1) | nxt()
    out = []
    n = len(int_cdf_rows)
    for i, row in enumerate(int_cdf_rows):
        top = len(row) - 2
        span = high - low + 1
        target = (((value - low + 1) << 16) - 1) // span
        target &= 0xFFFF
        # binary search exactly as published: largest index with cdf <= target among [0, top]
        left, right = 0, top + 1
        s = None
        while left + 1 < right:
            m = (left + right) // 2
            if row[m] < target:
                left = m
            elif row[m] > target:
                right = m
            else:
                s = m
                break
        if s is None:
            s = left
        out.append(s)
        if i == n - 1:
            break
        c_lo = row[s]
        c_hi = 65536 if s == top else row[s + 1]
        high = (low - 1 + ((span * c_hi) >> 16)) & FULL
        low = (low + ((span * c_lo) >> 16)) & FULL
        while True:
            if high < HALF or low >= HALF:
                pass
            elif low >= QUARTER and high < THREEQ:
                low -= QUARTER
                high -= QUARTER
                value -= QUARTER
            else:
                break
            low = (low << 1) & FULL
            high = ((high << 1) | 1) & FULL
            value = ((value << 1) | nxt()) & FULL
    return out


# ---- rANS (our hyper.b format) ------------------------------------------------------------------
RANS_L = 1 << 16


def _esc_bits(value, max_value):
    if value < 0:
        sign, m = 1, -value
    else:
        sign, m = 0, value - max_value + 1
    n = m.bit_length() - 1
    return [sign] + [0] * n + [1] + [(m >> i) & 1 for i in reversed(range(n))]


def rans_encode(symbols, cdf, cdf_len, offset, prec=16):
    """symbols [C][n] ints; returns bytes (state little-endian, then 16-bit words)."""
    C, n = len(symbols), len(symbols[0]) if symbols else 0
    x = RANS_L
    words = []

    def put(start, freq, p):
        nonlocal x
        x_max = ((RANS_L >> p) << 16) * freq
        while x >= x_max:
            words.append(x & 0xFFFF)
            x >>= 16
        x = ((x // freq) << p) + (x % freq) + start

    ops = []                                   # decoder order
    for i in range(n):
        for c in range(C):
            max_value = cdf_len[c] - 2
            v = symbols[c][i] - offset[c]
            if v < 0 or v >= max_value:
                ops.append((cdf[c][max_value], cdf[c][max_value + 1] - cdf[c][max_value], prec))
                ops.extend((b, 1, 1) for b in _esc_bits(v, max_value))
            else:
                ops.append((cdf[c][v], cdf[c][v + 1] - cdf[c][v], prec))
    for start, freq, p in reversed(ops):
        put(start, freq, p)
    out = bytearray(x.to_bytes(4, "little"))
    for w in reversed(words):
        out += w.to_bytes(2, "little")
    return bytes(out)
