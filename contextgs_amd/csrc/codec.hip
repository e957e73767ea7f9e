// Entropy coding for the bitstream container.
//
// 1. A 32-bit binary arithmetic coder over 16-bit integer CDFs: the algorithm of
//    the `torchac` wheel the reference calls (utils/encodings.py:108,138,157,178;
//    NOT in the mount, unpinned — restated from its public description, SURVEY
//    Appendix B): low/high registers, underflow ("pending") bits, MSB-first bit
//    packing, the top symbol's upper bound fixed at 2^16, one trailing
//    disambiguation bit.  The core is a __host__ __device__ template used by
//      * the table-driven host coder (torchac drop-in; Bernoulli mask stream), and
//      * the batched GPU Gaussian codec below.
// 2. The Gaussian codec of encoder_gaussian / decoder_gaussian
//    (utils/encodings.py:83-144) WITHOUT the [n_sym, L] float table: one WAVE per
//    1000-anchor chunk stream; the lanes evaluate the integer CDF entries the coder
//    needs (the two bounds of 64 symbols at a time when encoding, 16-entry windows of
//    four symbols at a time when decoding) with the same device erff on both sides, so
//    encode -> decode is bit-exact by construction and nothing crosses PCIe but the
//    bitstream.  Any number of streams (all levels and attributes) share a launch.
// 3. A range-ANS coder for the hyper-prior symbols (EntropyBottleneck.compress /
//    decompress; compressai is NOT in the mount): per-channel frequency tables,
//    escape symbol + Elias-gamma style bypass for out-of-support values.
#include <cstring>
#include <vector>
#include "cgs_internal.h"

#define AC_PRECISION 16
#define AC_TOP 0x10000u

// ---------------------------------------------------------------------------------
// bit I/O
// ---------------------------------------------------------------------------------
__host__ __device__ inline int clz32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return v ? __clz((int)v) : 32;
#else
    return v ? __builtin_clz(v) : 32;
#endif
}

struct BitWriter {
    uint8_t *p;
    uint8_t *end;
    uint64_t acc;     // pending output bits (< 8 between calls), MSB first
    int nacc;
    bool overflow;
    bool writer;      // wave-cooperative coders keep the state in every lane but let only one lane store
    __host__ __device__ void init(uint8_t *buf, size_t cap) { p = buf; end = buf + cap; acc = 0; nacc = 0; overflow = false; writer = true; }
    // append the low n bits of v (n <= 32), MSB first
    __host__ __device__ void put_bits(uint32_t v, int n) {
        acc = (acc << n) | (uint64_t)v;
        nacc += n;
        while (nacc >= 8) {
            const uint8_t byte = (uint8_t)(acc >> (nacc - 8));
            if (p < end) { if (writer) *p = byte; ++p; } else overflow = true;
            nacc -= 8;
        }
        acc &= 0xFFu;             // only the < 8 leftover bits matter
    }
    __host__ __device__ void put_with_pending(int bit, uint64_t &pending) {
        put_bits((uint32_t)bit, 1);
        while (pending > 0) {
            const int n = pending > 31 ? 31 : (int)pending;
            put_bits(bit ? 0u : ((1u << n) - 1u), n);
            pending -= (uint64_t)n;
        }
    }
    __host__ __device__ void flush() { if (nacc > 0) put_bits(0u, 8 - nacc); }
    __host__ __device__ size_t finish(uint8_t *base) { flush(); return (size_t)(p - base); }
};

struct BitReader {
    const uint8_t *buf;
    uint64_t len;
    uint64_t pos;     // bit position
    __host__ __device__ void init(const uint8_t *b, size_t n) { buf = b; len = (uint64_t)n; pos = 0; }
    // next n bits (1..32), MSB first; zeros past the end of the stream
    __host__ __device__ uint32_t get_bits(int n) {
        const uint64_t idx = pos >> 3;
        uint64_t w = 0;
        for (int t = 0; t < 5; ++t) w = (w << 8) | (idx + t < len ? (uint64_t)buf[idx + t] : 0ull);
        const int shift = 40 - (int)(pos & 7u) - n;
        pos += (uint64_t)n;
        return (uint32_t)((w >> shift) & ((1ull << n) - 1ull));
    }
};

// ---------------------------------------------------------------------------------
// arithmetic coder core
// ---------------------------------------------------------------------------------
// Renormalisation is done in bulk: the published bit-at-a-time loop always runs as (equal leading
// bits)* (underflow steps)*, so k = clz(low ^ high) settled bits are emitted at once (the first one
// followed by the pending run) and u = min(leading ones of low<<1, leading zeros of high<<1)
// underflow steps are applied at once.  Bit-for-bit the same stream as the per-bit loop
// (tests/test_codec.py pins it against the pure-Python bit-list oracle).
struct AcRenorm { int k, u; };

__host__ __device__ inline AcRenorm ac_renorm(uint32_t &low, uint32_t &high) {
    // Branch-free (the coders run this once per symbol on a lone wave, where a taken branch costs more than the
    // arithmetic it skips).  k in [0, 32]: 64-bit shifts make k = 32 (low == high) come out as low = 0,
    // high = 2^32 - 1.  After the k step low's top bit is 0 and high's is 1, so the u step's mask / or are
    // no-ops when u = 0.
    AcRenorm r;
    r.k = clz32(low ^ high);
    low = (uint32_t)((uint64_t)low << r.k);
    high = (uint32_t)(((uint64_t)high << r.k) | ((1ull << r.k) - 1ull));
    const uint32_t hs = high << 1;
    const int ones = clz32(~(low << 1)), zeros = hs ? clz32(hs) : 31;
    r.u = ones < zeros ? ones : zeros;
    low = (low << r.u) & 0x7FFFFFFFu;
    high = (high << r.u) | 0x80000000u | ((1u << r.u) - 1u);
    return r;
}

// Device encoder output: bits collect in a 64-bit accumulator and leave as whole big-endian dwords, so the
// per-symbol path is shift / or / add and one predictable branch instead of a byte loop.  The state is wave-uniform
// and EVERY lane stores the same dword to the same address (one write for the memory system): a `lane == 0`
// predicate around the store makes the compiler keep cursor and accumulator in vector registers behind exec-mask
// branches, which tripled the per-symbol instruction count.  The slot must be 4-byte aligned (the launchers hand
// out 8-byte aligned slots).
struct WaveBitWriter {
    uint32_t *p;
    uint8_t *end;
    uint64_t acc;     // the low nacc bits are pending output (nacc < 32 between calls); bits above them are stale
    int nacc;
    bool overflow;
    __host__ __device__ void init(uint8_t *buf, size_t cap) { p = (uint32_t *)buf; end = buf + cap; acc = 0; nacc = 0; overflow = false; }
    __host__ __device__ void put_bits(uint32_t v, int n) {          // low n bits of v (0 <= n <= 32, v < 2^n), MSB first
        acc = (acc << n) | (uint64_t)v;
        nacc += n;
        if (nacc >= 32) {
            const uint32_t w = (uint32_t)(acc >> (nacc - 32));
            nacc -= 32;
            if ((uint8_t *)(p + 1) <= end) { *p = __builtin_bswap32(w); ++p; } else overflow = true;
        }
    }
    __host__ __device__ void put_with_pending(int bit, uint64_t &pending) {
        put_bits((uint32_t)bit, 1);
        while (pending > 0) {
            const int n = pending > 31 ? 31 : (int)pending;
            put_bits(bit ? 0u : ((1u << n) - 1u), n);
            pending -= (uint64_t)n;
        }
    }
    // pad to a byte boundary with zeros and store the tail bytes; returns the stream length in bytes
    __host__ __device__ size_t finish(uint8_t *base) {
        uint8_t *q = (uint8_t *)p;
        const int nbytes = (nacc + 7) >> 3;
        const uint32_t tail = nacc ? (uint32_t)(acc << (32 - nacc)) : 0u;    // pending bits left-aligned in 32
        for (int t = 0; t < nbytes; ++t) {
            if (q < end) { *q = (uint8_t)(tail >> (24 - 8 * t)); ++q; } else overflow = true;
        }
        return (size_t)(q - base);
    }
};

template <class Writer>
struct AcEncoderT {
    uint32_t low, high;
    uint64_t pending;
    Writer out;
    __host__ __device__ void init(uint8_t *buf, size_t cap) { low = 0; high = 0xFFFFFFFFu; pending = 0; out.init(buf, cap); }
    __host__ __device__ void encode(uint32_t c_low, uint32_t c_high) {
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        high = (low - 1) + (uint32_t)((span * (uint64_t)c_high) >> AC_PRECISION);
        low = low + (uint32_t)((span * (uint64_t)c_low) >> AC_PRECISION);
        const uint32_t settled = low;            // its top k bits are the bits to emit
        const AcRenorm r = ac_renorm(low, high);
        const uint32_t bits = (uint32_t)(((uint64_t)settled << r.k) >> 32);     // top k bits of settled (0 if k = 0)
        if (pending == 0 || r.k == 0) {
            out.put_bits(bits, r.k);             // first bit + (no pending run) + the other k - 1 bits
        } else {
            out.put_with_pending((int)((bits >> (r.k - 1)) & 1u), pending);
            if (r.k > 1) out.put_bits(bits & ((1u << (r.k - 1)) - 1u), r.k - 1);
        }
        pending += (uint64_t)r.u;
    }
    __host__ __device__ void finish_bits() {
        ++pending;
        out.put_with_pending(low < 0x40000000u ? 0 : 1, pending);
    }
    __host__ __device__ size_t finish(uint8_t *base) {
        finish_bits();
        return out.finish(base);
    }
};
typedef AcEncoderT<BitWriter> AcEncoder;

template <class Reader>
struct AcDecoderT {
    uint32_t low, high, value;
    Reader in;
    __host__ __device__ void init(const uint8_t *buf, size_t len) {
        low = 0; high = 0xFFFFFFFFu;
        in.init(buf, len);
        value = in.get_bits(32);
    }
    __host__ __device__ uint32_t target() const {
        // floor(num / span) with num < 2^49, span <= 2^32: one fp64 division (exact operands, quotient < 2^17)
        // plus an integer fix-up instead of the ~150-instruction 64-bit integer division
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        const uint64_t num = ((uint64_t)value - (uint64_t)low + 1) * (uint64_t)AC_TOP - 1;
        uint64_t q = (uint64_t)((double)num / (double)span);
        while (q * span > num) --q;
        while ((q + 1) * span <= num) ++q;
        return (uint32_t)q & 0xFFFFu;
    }
    __host__ __device__ void consume(uint32_t c_low, uint32_t c_high) {
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        high = (low - 1) + (uint32_t)((span * (uint64_t)c_high) >> AC_PRECISION);
        low = low + (uint32_t)((span * (uint64_t)c_low) >> AC_PRECISION);
        const AcRenorm r = ac_renorm(low, high);
        if (r.k >= 32) { value = in.get_bits(32); }
        else if (r.k > 0) { value = (value << r.k) | in.get_bits(r.k); }
        // u underflow steps: each is value = ((value - 2^30) << 1) | bit  ==  ((value << u) ^ 2^31) | bits
        if (r.u > 0) value = ((value << r.u) ^ 0x80000000u) | in.get_bits(r.u);
    }
};

typedef AcDecoderT<BitReader> AcDecoder;

// Largest m in [0, max_sym] with cdf(m) <= target (cdf strictly increasing).
template <typename CdfFn>
__host__ __device__ inline int ac_search(CdfFn cdf, uint32_t target, int max_sym) {
    int left = 0, right = max_sym + 1;
    while (left + 1 < right) {
        const int m = (left + right) >> 1;
        const uint32_t v = cdf(m);
        if (v < target) left = m;
        else if (v > target) right = m;
        else return m;
    }
    return left;
}

// ---------------------------------------------------------------------------------
// host, table driven (torchac drop-in)
// ---------------------------------------------------------------------------------
extern "C" size_t cgs_ac_max_bytes(int64_t n_sym) { return (size_t)(n_sym > 0 ? n_sym : 0) * 2 + 16; }

// cdf_float [n_sym, Lp] -> uint16 [n_sym, Lp]: round(c * (2^16 - (Lp-1))) + j, wrapped to 16 bits
// (the published conversion; strictly increasing for any non-decreasing input row).
extern "C" int cgs_cdf_float_to_u16_host(const float *cdf, int64_t n_sym, int Lp, uint16_t *out) {
    if (!cdf || !out || Lp < 2 || n_sym < 0) { cgs_set_error("cdf_float_to_u16: bad args"); return CGS_ERR_ARG; }
    const float scale = (float)(65536 - (Lp - 1));
    for (int64_t i = 0; i < n_sym; ++i)
        for (int j = 0; j < Lp; ++j) {
            const float v = nearbyintf(cdf[i * Lp + j] * scale);
            out[i * Lp + j] = (uint16_t)((int32_t)v + j);
        }
    return CGS_OK;
}

extern "C" int cgs_ac_encode_table_host(const uint16_t *cdf, int Lp, const int16_t *sym, int64_t n_sym,
                                        uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!cdf || !sym || !out || !out_len || Lp < 2 || n_sym < 0) { cgs_set_error("ac_encode_table: bad args"); return CGS_ERR_ARG; }
    AcEncoder enc;
    enc.init(out, out_cap);
    const int max_sym = Lp - 2;
    for (int64_t i = 0; i < n_sym; ++i) {
        const int s = sym[i];
        if (s < 0 || s > max_sym) { cgs_set_error("ac_encode_table: symbol %d out of [0,%d] at %lld", s, max_sym, (long long)i); return CGS_ERR_BOUNDS; }
        const uint16_t *row = cdf + i * Lp;
        enc.encode(row[s], s == max_sym ? AC_TOP : (uint32_t)row[s + 1]);
    }
    *out_len = enc.finish(out);
    if (enc.out.overflow) { cgs_set_error("ac_encode_table: output buffer too small"); return CGS_ERR_WORKSPACE; }
    return CGS_OK;
}

extern "C" int cgs_ac_decode_table_host(const uint16_t *cdf, int Lp, int64_t n_sym, const uint8_t *in, size_t in_len,
                                        int16_t *sym_out) {
    if (!cdf || !sym_out || (!in && in_len) || Lp < 2 || n_sym < 0) { cgs_set_error("ac_decode_table: bad args"); return CGS_ERR_ARG; }
    AcDecoder dec;
    dec.init(in, in_len);
    const int max_sym = Lp - 2;
    for (int64_t i = 0; i < n_sym; ++i) {
        const uint16_t *row = cdf + i * Lp;
        const int s = ac_search([&](int m) { return (uint32_t)row[m]; }, dec.target(), max_sym);
        sym_out[i] = (int16_t)s;
        if (i == n_sym - 1) break;
        dec.consume(row[s], s == max_sym ? AC_TOP : (uint32_t)row[s + 1]);
    }
    return CGS_OK;
}

// Constant-CDF stream (the Bernoulli mask stream, utils/encodings.py:147-180): every symbol
// shares one row, so no [n,3] table is materialised.
// Two-symbol constant row, encoder side: the mask stream is ONE serial stream of N*K symbols (10 M at 1 M anchors) and
// the longest single job of conduct_encoding AND conduct_decoding (76 / 69 ms at 1 M anchors when this was written), so
// both directions get their own loops, written for the latency of the serial chain low/high -> span -> product ->
// renormalisation:
//   * one multiplication per symbol (symbol 1 keeps `high`, symbol 0 keeps `low`: c_high = 2^16 gives back the old high);
//   * no data-dependent branch on the output side: the settled bits, preceded by the pending run when there is one, are
//     assembled as ONE word of <= 56 bits with selects and appended to a 64-bit accumulator that is stored (8 bytes,
//     unconditionally) and advanced by whole bytes — whether a symbol settles bits is a coin flip the branch predictor
//     loses;
//   * symbols are validated in a vectorisable pre-pass.
// Same bits as the generic loop below (tests/test_codec.py pins both against the bit-list oracle).
struct HostBitSink {
    uint8_t *p, *end;
    uint64_t acc;         // the low nacc bits are not stored yet (nacc < 8 between calls)
    int nacc;
    bool overflow;
    void init(uint8_t *buf, size_t cap) { p = buf; end = buf + cap; acc = 0; nacc = 0; overflow = false; }
    inline void put(uint64_t v, int n) {             // n <= 56, v < 2^n
        acc = (acc << n) | v;
        nacc += n;
        const uint64_t w = __builtin_bswap64((acc << (63 - nacc)) << 1);    // pending bits left-aligned (nacc = 0: unused)
        if (p + 8 <= end) memcpy(p, &w, 8);
        else { for (int t = 0; t < (nacc >> 3); ++t) { if (p + t < end) p[t] = (uint8_t)(w >> (8 * t)); else overflow = true; } }
        p += nacc >> 3;
        nacc &= 7;
    }
    inline void put_run(int bit, uint64_t count) {   // `count` copies of `bit`
        while (count > 0) {
            const int n = count > 32 ? 32 : (int)count;
            put(bit ? ((1ull << n) - 1ull) : 0ull, n);
            count -= (uint64_t)n;
        }
    }
    size_t finish(uint8_t *base) {                   // zero-pad to a byte boundary
        if (nacc) put(0, 8 - nacc);
        if (p > end) overflow = true;
        return (size_t)(p - base);
    }
};

static int ac_encode_binary_host(const uint16_t *row, const int16_t *sym, int64_t n_sym, uint8_t *out, size_t out_cap,
                                 size_t *out_len) {
    int bad = 0;
    for (int64_t i = 0; i < n_sym; ++i) bad |= (sym[i] & ~1);
    if (bad) { cgs_set_error("ac_encode_const: symbol out of range"); return CGS_ERR_BOUNDS; }
    HostBitSink sink;
    sink.init(out, out_cap);
    const uint64_t c1 = row[1];
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint64_t pending = 0;
    for (int64_t i = 0; i < n_sym; ++i) {
        const uint64_t span = (uint64_t)(high - low) + 1;
        const uint32_t t = (uint32_t)((span * c1) >> AC_PRECISION);
        const bool one = sym[i] != 0;
        const uint32_t nh = (low - 1) + t, nl = low + t;
        high = one ? high : nh;
        low = one ? nl : low;
        const uint32_t settled = low;
        const AcRenorm r = ac_renorm(low, high);
        const int k = r.k;
        if (__builtin_expect(pending > 24, 0)) {     // a long underflow run: the plain path
            if (k) {
                const uint32_t bits = (uint32_t)(((uint64_t)settled << k) >> 32);
                const int b = (int)((bits >> (k - 1)) & 1u);
                sink.put((uint64_t)b, 1);
                sink.put_run(!b, pending);
                if (k > 1) sink.put(bits & ((1u << (k - 1)) - 1u), k - 1);
                pending = 0;
            }
            pending += (uint64_t)r.u;
            continue;
        }
        const bool has = k != 0;
        const int km1 = has ? k - 1 : 0;
        const uint64_t bits = ((uint64_t)settled << k) >> 32;                 // top k bits of settled (0 if k = 0)
        const uint64_t b = (bits >> km1) & 1ull;
        const int P = (int)pending;
        const uint64_t run = b ? 0ull : ((1ull << P) - 1ull);                 // the pending bits are the complement of b
        const uint64_t word = (b << (P + km1)) | (run << km1) | (bits & ((1ull << km1) - 1ull));
        sink.put(has ? word : 0ull, has ? k + P : 0);
        pending = (has ? 0ull : pending) + (uint64_t)r.u;
    }
    // the closing bits of AcEncoderT::finish_bits
    ++pending;
    const int b = low < 0x40000000u ? 0 : 1;
    sink.put((uint64_t)b, 1);
    sink.put_run(!b, pending);
    *out_len = sink.finish(out);
    if (sink.overflow) { cgs_set_error("ac_encode_const: output buffer too small"); return CGS_ERR_WORKSPACE; }
    return CGS_OK;
}

extern "C" int cgs_ac_encode_const_host(const uint16_t *row, int Lp, const int16_t *sym, int64_t n_sym, uint8_t *out,
                                        size_t out_cap, size_t *out_len) {
    if (!row || !sym || !out || !out_len || Lp < 2) { cgs_set_error("ac_encode_const: bad args"); return CGS_ERR_ARG; }
    if (Lp == 3 && row[0] == 0) return ac_encode_binary_host(row, sym, n_sym, out, out_cap, out_len);
    AcEncoder enc;
    enc.init(out, out_cap);
    const int max_sym = Lp - 2;
    for (int64_t i = 0; i < n_sym; ++i) {
        const int s = sym[i];
        if (s < 0 || s > max_sym) { cgs_set_error("ac_encode_const: symbol out of range"); return CGS_ERR_BOUNDS; }
        enc.encode(row[s], s == max_sym ? AC_TOP : (uint32_t)row[s + 1]);
    }
    *out_len = enc.finish(out);
    if (enc.out.overflow) { cgs_set_error("ac_encode_const: output buffer too small"); return CGS_ERR_WORKSPACE; }
    return CGS_OK;
}

// Two-symbol constant row: the mask stream is ONE serial stream of N*K symbols (10 M at 1 M anchors) and sits on the
// decoder's critical path, so it gets its own loop: no division (cdf[1] <= floor(num / span)  <=>  cdf[1] * span <= num),
// no search, a 64-bit bit buffer refilled eight bytes at a time.  Same symbols as the generic loop below.
struct HostBitSource {
    const uint8_t *buf;
    size_t len, pos;
    uint64_t bb;          // next bits, MSB first; the top nb are accounted for (bits below them are a harmless preview)
    int nb;
    void init(const uint8_t *b, size_t n) { buf = b; len = n; pos = 0; bb = 0; nb = 0; }
    inline void refill() {
        if (pos + 8 <= len) {
            uint64_t w;
            memcpy(&w, buf + pos, 8);
            bb |= __builtin_bswap64(w) >> nb;
            const int adv = (63 - nb) >> 3;
            pos += (size_t)adv;
            nb += adv * 8;
        } else {
            while (nb <= 56) {                       // zeros past the end of the stream
                bb |= (uint64_t)(pos < len ? buf[pos] : 0) << (56 - nb);
                ++pos;
                nb += 8;
            }
        }
    }
    inline uint32_t get_bits(int n) {               // 1 <= n <= 32
        if (nb < n) refill();
        const uint32_t r = (uint32_t)(bb >> (64 - n));
        bb <<= n;
        nb -= n;
        return r;
    }
};

static int ac_decode_binary_host(const uint16_t *row, int64_t n_sym, const uint8_t *in, size_t in_len, int16_t *sym_out) {
    HostBitSource src;
    src.init(in, in_len);
    uint32_t low = 0, high = 0xFFFFFFFFu, value = src.get_bits(32);
    const uint32_t c1 = row[1];
    for (int64_t i = 0; i < n_sym; ++i) {
        const uint64_t span = (uint64_t)(high - low) + 1;
        const uint64_t num = (((uint64_t)value - (uint64_t)low + 1) << AC_PRECISION) - 1;
        const uint64_t prod = (uint64_t)c1 * span;               // the one product both the test and the update need
        const bool one = prod <= num;
        sym_out[i] = (int16_t)one;
        if (i == n_sym - 1) break;
        // symbol 1 = [c1, 2^16): high stays (span * 2^16 >> 16 = span), low moves up; symbol 0 = [0, c1): low stays.
        // Selected with masks, and the bit refill below takes n = 0 in its stride: which symbol it was and whether it
        // settles bits are coin flips, so nothing on this path may be a branch.
        const uint32_t t = (uint32_t)(prod >> AC_PRECISION);
        const uint32_t m1 = 0u - (uint32_t)one;
        high = (high & m1) | (((low - 1) + t) & ~m1);
        low = low + (t & m1);
        const AcRenorm r = ac_renorm(low, high);
        const int n = r.k + r.u;
        if (__builtin_expect(n > 32, 0)) {
            value = r.k >= 32 ? src.get_bits(32) : ((value << r.k) | src.get_bits(r.k));
            value = ((value << r.u) ^ 0x80000000u) | src.get_bits(r.u);
            continue;
        }
        if (src.nb < n) src.refill();
        const uint32_t fresh = (uint32_t)((src.bb >> 1) >> (63 - n));        // the next n bits (n = 0: none)
        src.bb <<= n;
        src.nb -= n;
        value = (uint32_t)(((uint64_t)value << n) | (uint64_t)fresh) ^ ((uint32_t)(r.u != 0) << 31);
    }
    return CGS_OK;
}

extern "C" int cgs_ac_decode_const_host(const uint16_t *row, int Lp, int64_t n_sym, const uint8_t *in, size_t in_len,
                                        int16_t *sym_out) {
    if (!row || !sym_out || Lp < 2) { cgs_set_error("ac_decode_const: bad args"); return CGS_ERR_ARG; }
    if (Lp == 3 && row[0] == 0) return ac_decode_binary_host(row, n_sym, in, in_len, sym_out);
    AcDecoder dec;
    dec.init(in, in_len);
    const int max_sym = Lp - 2;
    for (int64_t i = 0; i < n_sym; ++i) {
        const int s = ac_search([&](int m) { return (uint32_t)row[m]; }, dec.target(), max_sym);
        sym_out[i] = (int16_t)s;
        if (i == n_sym - 1) break;
        dec.consume(row[s], s == max_sym ? AC_TOP : (uint32_t)row[s + 1]);
    }
    return CGS_OK;
}

// ---------------------------------------------------------------------------------
// GPU Gaussian codec
// ---------------------------------------------------------------------------------
#define SQRT2F 1.4142135623730951f

// Integer CDF entry j of a symbol with (mean, scale, Q) over the grid [min_v .. max_v + 1]:
// the reference's table entry lower[i, j] = Normal(mean, scale).cdf((min_v + j - 0.5) * Q)
// (utils/encodings.py:91-97) pushed through the float -> int16 conversion above.
__device__ __forceinline__ uint32_t gaussian_cdf_int(int j, int min_v, float norm, float mean, float inv_scale,
                                                     float q) {
    const float sample = ((float)(min_v + j) - 0.5f) * q;
    // torch divides a device tensor by the host scalar sqrt(2) as a multiplication by its reciprocal
    const float z = (sample - mean) * inv_scale * (1.f / SQRT2F);
    const float c = 0.5f * (1.f + erff(z));
    return (uint32_t)((int32_t)rintf(c * norm) + j) & 0xFFFFu;
}

// per-stream min/max of round(x/Q)
__global__ void __launch_bounds__(256)
    stream_minmax_kernel(const float *__restrict__ x, const float *__restrict__ Q, int64_t q_div,
                         const int64_t *__restrict__ stream_off, int n_streams, int32_t *__restrict__ min_out,
                         int32_t *__restrict__ max_out) {
    const int s = blockIdx.x;
    if (s >= n_streams) return;
    const int64_t b = stream_off[s], e = stream_off[s + 1];
    int lo = INT32_MAX, hi = INT32_MIN;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) {
        const int v = (int)rintf(x[i] / Q[i / q_div]);
        lo = min(lo, v);
        hi = max(hi, v);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, __shfl_xor(lo, d, 64));
        hi = max(hi, __shfl_xor(hi, d, 64));
    }
    __shared__ int slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = min(min(slo[0], slo[1]), min(slo[2], slo[3]));
        hi = max(max(shi[0], shi[1]), max(shi[2], shi[3]));
        if (e <= b) { lo = 0; hi = 0; }
        min_out[s] = lo;
        max_out[s] = hi;
    }
}

__device__ __forceinline__ float bcast_f(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}
__device__ __forceinline__ uint32_t bcast_u(uint32_t v, int src_lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane);
}

// ONE WAVE PER STREAM.  The 64 lanes evaluate the integer CDF entries of 64 consecutive symbols in
// parallel (coalesced loads of x / mean / scale, 2 erff per lane), then the range coder — whose state is
// wave-uniform — consumes them one by one via v_readlane; only lane 0 writes bytes.  All chunk streams of a
// level/attribute run concurrently, one wave each, instead of one serial LANE each (which left 63/64 of
// every wave and most of the chip idle: 190 ms per launch at 1 M anchors).
__global__ void __launch_bounds__(64)
    gaussian_encode_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                           const float *__restrict__ scale, const float *__restrict__ Q, int64_t q_div,
                           const int64_t *__restrict__ stream_off, int n_streams,
                           const int32_t *__restrict__ min_v, const int32_t *__restrict__ max_v,
                           uint8_t *__restrict__ out, const int64_t *__restrict__ out_off,
                           uint32_t *__restrict__ out_len, int32_t *__restrict__ status) {
    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    if (s >= n_streams) return;
    const int64_t b = stream_off[s], e = stream_off[s + 1];
    uint8_t *base = out + out_off[s];
    AcEncoderT<WaveBitWriter> enc;
    enc.init(base, (size_t)(out_off[s + 1] - out_off[s]));
    const int lo = min_v[s];
    const int Lp = max_v[s] - lo + 2;
    const int max_sym = Lp - 2;
    const float norm = (float)(65536 - (Lp - 1));
    bool bad = Lp > 65536;
    for (int64_t i0 = b; i0 < e && !bad; i0 += 64) {
        const int64_t i = i0 + lane;
        uint32_t c_low = 0, c_high = 1;
        bool my_bad = false;
        if (i < e) {
            const float q = Q[i / q_div];
            const int sym = (int)rintf(x[i] / q) - lo;
            if (sym < 0 || sym > max_sym) {
                my_bad = true;
            } else {
                const float inv = 1.f / scale[i];
                const float m = mean[i];
                c_low = gaussian_cdf_int(sym, lo, norm, m, inv, q);
                c_high = sym == max_sym ? AC_TOP : gaussian_cdf_int(sym + 1, lo, norm, m, inv, q);
            }
        }
        if (__ballot(my_bad) != 0ull) { bad = true; break; }
        const int cnt = (int)min((int64_t)64, e - i0);
        for (int j = 0; j < cnt; ++j) enc.encode(bcast_u(c_low, j), bcast_u(c_high, j));
    }
    const uint32_t len = (uint32_t)enc.finish(base);
    if (lane == 0) {
        out_len[s] = len;
        if (bad) atomicMax(status, 1);
        if (enc.out.overflow) atomicMax(status, 2);
    }
}

// Device decoder input: MSB-first bit source with a wave-uniform 64-bit bit buffer fed from a 256-byte register
// window of the stream (lane l = big-endian dword l of the window): get_bits is shift / subtract, a refill is one
// v_readlane, and memory is touched once per 256 stream bytes.  Zeros past the end of the stream.
struct WaveBitSource {
    const uint8_t *buf;
    uint64_t len, next;      // stream bytes; byte offset of the next window
    uint32_t win;
    int d;                   // next dword of the window to shift in
    uint64_t bb;             // bit buffer, the top nb bits are valid, the rest are zero
    int nb;
    __device__ void fill() {
        const uint64_t o = next + 4ull * (threadIdx.x & 63);
        uint32_t w = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) w = (w << 8) | (o + t < len ? (uint32_t)buf[o + t] : 0u);
        win = w;
        next += 256;
        d = 0;
    }
    __device__ void init(const uint8_t *b, size_t n) { buf = b; len = (uint64_t)n; next = 0; bb = 0; nb = 0; fill(); }
    __device__ uint32_t get_bits(int n) {           // 1 <= n <= 32
        if (nb < 32) {
            if (d == 64) fill();
            bb |= (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)win, d) << (32 - nb);
            ++d;
            nb += 32;
        }
        const uint32_t r = (uint32_t)(bb >> (64 - n));
        bb <<= n;
        nb -= n;
        return r;
    }
};

struct WaveAcDecoder {
    uint32_t low, high, value;
    WaveBitSource in;
    __device__ void init(const uint8_t *buf, size_t len) {
        low = 0; high = 0xFFFFFFFFu;
        in.init(buf, len);
        value = in.get_bits(32);
    }
    // cdf <= target  <=>  cdf * span <= num  (target = floor(num / span), AcDecoderT::target): the per-lane test
    // is one 64-bit multiply-add and a compare, no division on the per-symbol path
    __device__ uint64_t num() const { return (((uint64_t)value - (uint64_t)low + 1) << AC_PRECISION) - 1; }
    __device__ uint32_t span_m1() const { return high - low; }
    __device__ void consume(uint32_t c_low, uint32_t c_high) {
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        high = (low - 1) + (uint32_t)((span * (uint64_t)c_high) >> AC_PRECISION);
        low = low + (uint32_t)((span * (uint64_t)c_low) >> AC_PRECISION);
        const AcRenorm r = ac_renorm(low, high);
        const int n = r.k + r.u;
        if (n == 0) return;
        if (n <= 32) {
            // k settled bits then u underflow steps (each value = ((value - 2^30) << 1) | bit) in one read:
            // ((value << k | bits_k) << u ^ 2^31) | bits_u  ==  (value << n | bits_n) ^ (u ? 2^31 : 0)
            value = (uint32_t)(((uint64_t)value << n) | (uint64_t)in.get_bits(n));
            if (r.u > 0) value ^= 0x80000000u;
        } else {
            value = r.k >= 32 ? in.get_bits(32) : ((value << r.k) | in.get_bits(r.k));
            value = ((value << r.u) ^ 0x80000000u) | in.get_bits(r.u);
        }
    }
};

__device__ __forceinline__ bool cdf_le_target(uint32_t cdf, uint32_t span_m1, uint64_t num) {
    return (uint64_t)cdf * (uint64_t)span_m1 + (uint64_t)cdf <= num;
}

// Decoder, ONE WAVE PER STREAM.  The integer CDF entries a symbol needs do not depend on the coder state, so they
// are evaluated ahead of it: the four 16-lane rows of the wave evaluate 16-entry windows (centred on the Gaussian's
// mean, where the mass is) of FOUR consecutive symbols with one erff, then the coder walks the four symbols; per
// symbol a ballot over `cdf * span <= num` replaces the binary search.  A symbol whose target falls outside its
// window takes the 64-wide search (all lanes on that one symbol, moving window).  Distribution parameters are
// fetched 64 symbols at a time, coalesced; decoded symbols collect in a register (v_writelane) and are dequantised
// and stored 64 at a time.
__global__ void __launch_bounds__(64)
    gaussian_decode_kernel(const float *__restrict__ mean, const float *__restrict__ scale,
                           const float *__restrict__ Q, int64_t q_div, const int64_t *__restrict__ stream_off,
                           int n_streams, const int32_t *__restrict__ min_v, const int32_t *__restrict__ max_v,
                           const uint8_t *__restrict__ in, const int64_t *__restrict__ in_off,
                           float *__restrict__ x_out) {
    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    if (s >= n_streams) return;
    const int64_t b = stream_off[s], e = stream_off[s + 1];
    WaveAcDecoder dec;
    dec.init(in + in_off[s], (size_t)(in_off[s + 1] - in_off[s]));
    const int lo = min_v[s];
    const int Lp = max_v[s] - lo + 2;
    const int max_sym = Lp - 2;
    const float norm = (float)(65536 - (Lp - 1));
    const int row = lane >> 4, col = lane & 15;
    for (int64_t i0 = b; i0 < e; i0 += 64) {
        const int64_t i = i0 + lane;
        float q_l = 1.f, m_l = 0.f, inv_l = 1.f;
        if (i < e) {
            q_l = Q[i / q_div];
            m_l = mean[i];
            inv_l = 1.f / scale[i];
        }
        const int mid_l = (int)rintf(m_l / q_l) - lo;        // symbol index nearest to the mean
        int sym_l = 0;
        const int cnt = (int)min((int64_t)64, e - i0);
        const int last_j = (int)min((int64_t)64, e - 1 - i0);   // the stream's final symbol is not consumed
        for (int jb = 0; jb < cnt; jb += 4) {
            const int src = jb + row;
            const float q4 = __shfl(q_l, src), m4 = __shfl(m_l, src), inv4 = __shfl(inv_l, src);
            const int cb4 = max(0, min(__shfl(mid_l, src) - 8, max_sym - 15));
            const int cand4 = cb4 + col;
            const uint32_t cdf4 = gaussian_cdf_int(cand4, lo, norm, m4, inv4, q4);
            const bool ok4 = cand4 <= max_sym;
            const int jn = min(4, cnt - jb);
            for (int r = 0; r < jn; ++r) {
                const int j = jb + r;
                const uint64_t num = dec.num();
                const uint32_t sm1 = dec.span_m1();
                const uint32_t rowbits = (uint32_t)(__ballot(cdf_le_target(cdf4, sm1, num) & ok4) >> (16 * r)) & 0xFFFFu;
                const int n_le = __builtin_popcount(rowbits);
                const int cbs = __builtin_amdgcn_readlane(cb4, 16 * r);
                int sym;
                uint32_t c_low, c_high;
                if ((uint32_t)(n_le - 1) < 15u) {
                    // 1..15 lanes pass and they form a prefix of the row (the CDF is strictly increasing): the symbol
                    // is the last of them and both of its bounds are in the window.  0 or 16 passing lanes (target
                    // outside the window, or the window touching an end of the alphabet) take the search below.
                    const int rel = n_le - 1;
                    sym = cbs + rel;
                    c_low = bcast_u(cdf4, 16 * r + rel);
                    c_high = sym >= max_sym ? AC_TOP : bcast_u(cdf4, 16 * r + rel + 1);
                } else {
                    const float q = bcast_f(q_l, j), m = bcast_f(m_l, j), inv = bcast_f(inv_l, j);
                    int cb = n_le == 0 ? cbs - 64 : cbs + 15;
                    cb = max(0, min(cb, max_sym - 63));
                    uint32_t cdf_l = 0;
                    for (;;) {
                        const int cand = cb + lane;
                        cdf_l = gaussian_cdf_int(cand, lo, norm, m, inv, q);
                        const bool le = cand <= max_sym && cdf_le_target(cdf_l, sm1, num);
                        const int n64 = __builtin_popcountll(__ballot(le));
                        if (n64 == 0) {
                            if (cb == 0) { sym = 0; break; }
                            cb = max(0, cb - 64);             // everything in the window is above the target
                        } else if (n64 == 64 && cb + 64 <= max_sym) {
                            cb += 63;                         // keep the last passing candidate in the next window
                        } else {
                            sym = cb + n64 - 1;
                            break;
                        }
                    }
                    const int rel = __builtin_amdgcn_readfirstlane(sym - cb);
                    if (rel >= 0 && rel < 63) {
                        c_low = bcast_u(cdf_l, rel);
                        c_high = sym == max_sym ? AC_TOP : bcast_u(cdf_l, rel + 1);
                    } else {
                        c_low = gaussian_cdf_int(sym, lo, norm, m, inv, q);
                        c_high = sym == max_sym ? AC_TOP : gaussian_cdf_int(sym + 1, lo, norm, m, inv, q);
                    }
                }
                sym_l = lane == j ? sym : sym_l;
                if (j != last_j) dec.consume(c_low, c_high);
            }
        }
        if (i < e) x_out[i] = (float)(sym_l + lo) * q_l;
    }
}

// test hook: the full integer table of one stream, [n, Lp] uint16 (what the reference ships over PCIe as floats)
__global__ void __launch_bounds__(256)
    gaussian_table_kernel(const float *__restrict__ mean, const float *__restrict__ scale, const float *__restrict__ Q,
                          int64_t q_div, int64_t n, int min_v, int Lp, uint16_t *__restrict__ table) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * Lp) return;
    const int64_t i = idx / Lp;
    const int j = (int)(idx % Lp);
    table[idx] = (uint16_t)gaussian_cdf_int(j, min_v, (float)(65536 - (Lp - 1)), mean[i], 1.f / scale[i], Q[i / q_div]);
}

extern "C" int cgs_gaussian_stream_minmax(const float *x, const float *Q, int64_t q_div, const int64_t *stream_off,
                                          int n_streams, int32_t *min_out, int32_t *max_out, void *stream) {
    if (n_streams < 0 || q_div < 1) { cgs_set_error("stream_minmax: bad args"); return CGS_ERR_ARG; }
    if (n_streams == 0) return CGS_OK;
    hipLaunchKernelGGL(stream_minmax_kernel, dim3(n_streams), dim3(256), 0, (hipStream_t)stream, x, Q, q_div, stream_off,
                       n_streams, min_out, max_out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_gaussian_ac_encode(const float *x, const float *mean, const float *scale, const float *Q,
                                      int64_t q_div, const int64_t *stream_off, int n_streams, const int32_t *min_v,
                                      const int32_t *max_v, uint8_t *out, const int64_t *out_off, uint32_t *out_len,
                                      int32_t *status, void *stream) {
    if (n_streams < 0 || q_div < 1) { cgs_set_error("gaussian_ac_encode: bad args"); return CGS_ERR_ARG; }
    if (n_streams == 0) return CGS_OK;
    hipLaunchKernelGGL(gaussian_encode_kernel, dim3(n_streams), dim3(64), 0, (hipStream_t)stream, x, mean,
                       scale, Q, q_div, stream_off, n_streams, min_v, max_v, out, out_off, out_len, status);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_gaussian_ac_decode(const float *mean, const float *scale, const float *Q, int64_t q_div,
                                      const int64_t *stream_off, int n_streams, const int32_t *min_v,
                                      const int32_t *max_v, const uint8_t *in, const int64_t *in_off, float *x_out,
                                      void *stream) {
    if (n_streams < 0 || q_div < 1) { cgs_set_error("gaussian_ac_decode: bad args"); return CGS_ERR_ARG; }
    if (n_streams == 0) return CGS_OK;
    hipLaunchKernelGGL(gaussian_decode_kernel, dim3(n_streams), dim3(64), 0, (hipStream_t)stream, mean,
                       scale, Q, q_div, stream_off, n_streams, min_v, max_v, in, in_off, x_out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_gaussian_cdf_table(const float *mean, const float *scale, const float *Q, int64_t q_div, int64_t n,
                                      int min_v, int max_v, uint16_t *table, void *stream) {
    const int Lp = max_v - min_v + 2;
    if (n < 0 || Lp < 2 || q_div < 1) { cgs_set_error("gaussian_cdf_table: bad args"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    const int64_t tot = n * Lp;
    hipLaunchKernelGGL(gaussian_table_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       mean, scale, Q, q_div, n, min_v, Lp, table);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// ---------------------------------------------------------------------------------
// Bernoulli chunk streams on the device (container version 2, masks.b)
// ---------------------------------------------------------------------------------
// The reference codes the N*K offset masks as ONE arithmetic-coded stream with a constant two-symbol CDF
// (scene/gaussian_model.py:1265-1269,1348-1353; utils/encodings.py:147-180): a serial chain of 10 M symbols at
// 1 M anchors that no device can parallelise and that the version-1 container therefore codes on a host thread.
// Version 2 of the container cuts the SAME symbol sequence into chunk streams (as the reference itself does for every
// other attribute) and codes each with the same coder core: one wave per stream, the 64 symbols of a step fetched by
// one coalesced load + ballot, the wave-uniform coder walking the ballot's bits on the scalar unit.  Stream s holds
// exactly the bytes cgs_ac_encode_const_host produces for its symbols (tests/test_codec_gpu.py).
__global__ void __launch_bounds__(64)
    bernoulli_encode_kernel(const float *__restrict__ sym01, uint32_t c1, const int64_t *__restrict__ stream_off,
                            int n_streams, uint8_t *__restrict__ out, const int64_t *__restrict__ out_off,
                            uint32_t *__restrict__ out_len, int32_t *__restrict__ status) {
    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    if (s >= n_streams) return;
    const int64_t b = stream_off[s], e = stream_off[s + 1];
    uint8_t *base = out + out_off[s];
    AcEncoderT<WaveBitWriter> enc;
    enc.init(base, (size_t)(out_off[s + 1] - out_off[s]));
    bool bad = false;
    for (int64_t i0 = b; i0 < e; i0 += 64) {
        const int64_t i = i0 + lane;
        const float v = i < e ? sym01[i] : 0.f;
        if (__ballot(v != 0.f && v != 1.f) != 0ull) { bad = true; break; }
        const uint64_t ones = __ballot(v != 0.f);
        const int cnt = (int)min((int64_t)64, e - i0);
        for (int j = 0; j < cnt; ++j) {
            const bool one = (ones >> j) & 1ull;
            enc.encode(one ? c1 : 0u, one ? AC_TOP : c1);
        }
    }
    const uint32_t len = (uint32_t)enc.finish(base);
    if (lane == 0) {
        out_len[s] = len;
        if (bad) atomicMax(status, 1);
        if (enc.out.overflow) atomicMax(status, 2);
    }
}

__global__ void __launch_bounds__(64)
    bernoulli_decode_kernel(uint32_t c1, const int64_t *__restrict__ stream_off, int n_streams,
                            const uint8_t *__restrict__ in, const int64_t *__restrict__ in_off,
                            float *__restrict__ sym_out) {
    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    if (s >= n_streams) return;
    const int64_t b = stream_off[s], e = stream_off[s + 1];
    WaveAcDecoder dec;
    dec.init(in + in_off[s], (size_t)(in_off[s + 1] - in_off[s]));
    for (int64_t i0 = b; i0 < e; i0 += 64) {
        const int cnt = (int)min((int64_t)64, e - i0);
        const int last_j = (int)min((int64_t)64, e - 1 - i0);   // the stream's final symbol is not consumed
        uint64_t ones = 0;
        for (int j = 0; j < cnt; ++j) {
            const bool one = cdf_le_target(c1, dec.span_m1(), dec.num());
            ones |= (uint64_t)one << j;
            if (j != last_j) dec.consume(one ? c1 : 0u, one ? AC_TOP : c1);
        }
        const int64_t i = i0 + lane;
        if (i < e) sym_out[i] = (ones >> lane) & 1ull ? 1.f : 0.f;
    }
}

extern "C" int cgs_bernoulli_ac_encode(const float *sym01, uint32_t c1, const int64_t *stream_off, int n_streams,
                                       uint8_t *out, const int64_t *out_off, uint32_t *out_len, int32_t *status,
                                       void *stream) {
    if (n_streams < 0 || c1 == 0 || c1 >= AC_TOP) { cgs_set_error("bernoulli_ac_encode: bad args"); return CGS_ERR_ARG; }
    if (n_streams == 0) return CGS_OK;
    hipLaunchKernelGGL(bernoulli_encode_kernel, dim3(n_streams), dim3(64), 0, (hipStream_t)stream, sym01, c1, stream_off,
                       n_streams, out, out_off, out_len, status);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_bernoulli_ac_decode(uint32_t c1, const int64_t *stream_off, int n_streams, const uint8_t *in,
                                       const int64_t *in_off, float *sym_out, void *stream) {
    if (n_streams < 0 || c1 == 0 || c1 >= AC_TOP) { cgs_set_error("bernoulli_ac_decode: bad args"); return CGS_ERR_ARG; }
    if (n_streams == 0) return CGS_OK;
    hipLaunchKernelGGL(bernoulli_decode_kernel, dim3(n_streams), dim3(64), 0, (hipStream_t)stream, c1, stream_off,
                       n_streams, in, in_off, sym_out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// ---------------------------------------------------------------------------------
// Lane-parallel Gaussian codec (container version 2: feat / scaling / offsets)
// ---------------------------------------------------------------------------------
// The reference's container cuts every attribute into 1000-anchor chunk streams (scene/gaussian_model.py:1071,1192-1238):
// few, long streams (50 000 symbols for a feature chunk).  A stream is a serial chain, so the kernels above give each one
// a WAVE whose 64 lanes only prepare operands for a coder that runs on the scalar unit: ~130 scalar instructions per symbol,
// one scalar unit per CU — 62 M symbols take ~17 ms (encode) / ~22 ms (decode, three dependent launches) at 1 M anchors
// however short the streams are cut.  Version 2 re-cuts the SAME symbol sequence for the machine: a BLOCK of up to
// 64 * L consecutive symbols is coded by one wave as 64 INTERLEAVED lane streams (lane l codes symbols l, l + 64, l + 128,
// ... of the block), each lane running its own instance of the same arithmetic coder in vector registers.  Loads of
// x / mean / scale stay coalesced (step t reads elements 64 t + lane), the coder arithmetic is 64-wide, and a block costs
// L serial steps instead of 64 L.  Block layout in the file: 64 little-endian uint16 byte lengths, then the 64 lane streams
// back to back; min / max of the block's symbols (CDF normalisation) and the block's byte length travel in meta.b like the
// chunk streams' do.  Lane stream l of a block is byte for byte what AcEncoderT / the table coder produce for the block's
// symbols l, l + 64, ... (tests/test_codec_gpu.py).
#define LANES_HDR 128
#ifndef LANES_AHEAD
#define LANES_AHEAD 3      // steps of operand prefetch in the lane kernels
#endif

// element -> step-size index (i / q_div) of a lane that walks i, i + 64, i + 128, ...: one 64-bit division at the start, then
// an add and a compare per step (a variable 64-bit division is ~100 instructions, and a lane step is a serial chain)
struct LaneQIndex {
    int64_t quot, rem, div;
    int dq, dr;
    __device__ void init(int64_t i, int64_t q_div) {
        quot = i / q_div; rem = i - quot * q_div; div = q_div;
        dq = q_div > 64 ? 0 : (int)(64 / q_div);
        dr = q_div > 64 ? 64 : (int)(64 % q_div);
    }
    __device__ void step() {
        quot += dq; rem += dr;
        if (rem >= div) { rem -= div; ++quot; }
    }
};

struct LaneBitSource {
    const uint8_t *p;
    uint64_t bb;      // next bits, MSB first; the top nb are valid, the rest zero
    uint32_t ahead;   // the dword at p, fetched when the previous one was consumed: a refill never waits for its own load
    int nb;
    __device__ void init(const uint8_t *b) { p = b; bb = 0; nb = 0; __builtin_memcpy(&ahead, p, 4); }
    __device__ uint32_t get_bits(int n) {           // 1 <= n <= 32; may read up to 8 bytes past the lane stream (any bits do)
        if (nb < n) {
            const uint32_t w = ahead;
            p += 4;
            __builtin_memcpy(&ahead, p, 4);
            bb |= (uint64_t)__builtin_bswap32(w) << (32 - nb);
            nb += 32;
        }
        const uint32_t r = (uint32_t)(bb >> (64 - n));
        bb <<= n;
        nb -= n;
        return r;
    }
};

struct LaneAcDecoder {
    uint32_t low, high, value;
    LaneBitSource in;
    __device__ void init(const uint8_t *buf) {
        low = 0; high = 0xFFFFFFFFu;
        in.init(buf);
        value = in.get_bits(32);
    }
    __device__ uint64_t num() const { return (((uint64_t)value - (uint64_t)low + 1) << AC_PRECISION) - 1; }
    __device__ uint32_t span_m1() const { return high - low; }
    __device__ void consume(uint32_t c_low, uint32_t c_high) {
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        high = (low - 1) + (uint32_t)((span * (uint64_t)c_high) >> AC_PRECISION);
        low = low + (uint32_t)((span * (uint64_t)c_low) >> AC_PRECISION);
        const AcRenorm r = ac_renorm(low, high);
        const int n = r.k + r.u;
        if (n == 0) return;
        if (n <= 32) {
            value = (uint32_t)(((uint64_t)value << n) | (uint64_t)in.get_bits(n));
            if (r.u > 0) value ^= 0x80000000u;
        } else {
            value = r.k >= 32 ? in.get_bits(32) : ((value << r.k) | in.get_bits(r.k));
            value = ((value << r.u) ^ 0x80000000u) | in.get_bits(r.u);
        }
    }
};

// bytes of the worst-case slot of one lane stream of a block with nsym symbols at bps bytes per symbol (multiple of 8)
__host__ __device__ inline int64_t lanes_slot_bytes(int64_t nsym, int bps = 2) { return (((nsym + 63) / 64) * bps + 16 + 7) / 8 * 8; }

extern "C" size_t cgs_lanes_block_slot_bytes(int64_t nsym, int bytes_per_symbol) {
    return (size_t)(LANES_HDR + 64 * lanes_slot_bytes(nsym > 0 ? nsym : 0, bytes_per_symbol));
}

// ONE WAVE PER BLOCK, one coder per lane.  out + out_off[b]: the block's worst-case region (cgs_lanes_block_slot_bytes): the
// header's 64 lengths, then 64 lane slots of lanes_slot_bytes each.  out_len[b] = LANES_HDR + sum of the lane lengths.
__global__ void __launch_bounds__(64)
    gaussian_encode_lanes_kernel(const float *__restrict__ x, const float *__restrict__ mean, const float *__restrict__ scale,
                                 const float *__restrict__ Q, int64_t q_div, const int64_t *__restrict__ blk_off, int n_blocks,
                                 const int32_t *__restrict__ min_v, const int32_t *__restrict__ max_v,
                                 uint8_t *__restrict__ out, const int64_t *__restrict__ out_off,
                                 uint32_t *__restrict__ out_len, int32_t *__restrict__ status) {
    const int blk = blockIdx.x, lane = threadIdx.x;
    if (blk >= n_blocks) return;
    const int64_t b = blk_off[blk], e = blk_off[blk + 1];
    const int64_t slot = lanes_slot_bytes(e - b);
    uint8_t *base = out + out_off[blk];
    uint8_t *mine = base + LANES_HDR + lane * slot;
    AcEncoderT<WaveBitWriter> enc;                 // a lane-private instance: every member lives in this lane's registers
    enc.init(mine, (size_t)slot);
    const int lo = min_v[blk];
    const int Lp = max_v[blk] - lo + 2;
    const int max_sym = Lp - 2;
    const float norm = (float)(65536 - (Lp - 1));
    bool bad = Lp > 65536;
    float q_n[LANES_AHEAD], x_n[LANES_AHEAD], m_n[LANES_AHEAD], sc_n[LANES_AHEAD];   // operands fetched LANES_AHEAD steps ahead
    LaneQIndex qix;
    qix.init(b + lane, q_div);
#pragma unroll
    for (int a = 0; a < LANES_AHEAD; ++a) {
        const int64_t j = b + lane + 64 * a;
        q_n[a] = 1.f; x_n[a] = 0.f; m_n[a] = 0.f; sc_n[a] = 1.f;
        if (j < e) { q_n[a] = Q[qix.quot]; x_n[a] = x[j]; m_n[a] = mean[j]; sc_n[a] = scale[j]; }
        qix.step();
    }
    for (int64_t i = b + lane; i < e && !bad; i += 64) {
        const float q = q_n[0], xv = x_n[0], m = m_n[0], inv = 1.f / sc_n[0];
#pragma unroll
        for (int a = 0; a + 1 < LANES_AHEAD; ++a) { q_n[a] = q_n[a + 1]; x_n[a] = x_n[a + 1]; m_n[a] = m_n[a + 1]; sc_n[a] = sc_n[a + 1]; }
        {
            const int64_t j = i + 64 * LANES_AHEAD;
            if (j < e) { q_n[LANES_AHEAD - 1] = Q[qix.quot]; x_n[LANES_AHEAD - 1] = x[j]; m_n[LANES_AHEAD - 1] = mean[j];
                         sc_n[LANES_AHEAD - 1] = scale[j]; }
            qix.step();
        }
        const int sym = (int)rintf(xv / q) - lo;
        if (sym < 0 || sym > max_sym) { bad = true; break; }
        const uint32_t c_low = gaussian_cdf_int(sym, lo, norm, m, inv, q);
        const uint32_t c_high = sym == max_sym ? AC_TOP : gaussian_cdf_int(sym + 1, lo, norm, m, inv, q);
        enc.encode(c_low, c_high);
    }
    const uint32_t len = (uint32_t)enc.finish(mine);
    ((uint16_t *)base)[lane] = (uint16_t)len;
    uint32_t tot = len;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d, 64);
    if (__ballot(bad) != 0ull && lane == 0) atomicMax(status, 1);
    if (__ballot(enc.out.overflow || len > 65535u) != 0ull && lane == 0) atomicMax(status, 2);
    if (lane == 0) out_len[blk] = LANES_HDR + tot;
}

// Pack the blocks: header + the 64 lane streams back to back at dst + dst_off[b] (the file layout).
__global__ void __launch_bounds__(256)
    lanes_compact_kernel(const uint8_t *__restrict__ src, const int64_t *__restrict__ src_off, const int64_t *__restrict__ blk_off,
                         const int64_t *__restrict__ dst_off, int n_blocks, uint8_t *__restrict__ dst, int bps) {
    const int blk = blockIdx.x, tid = threadIdx.x;
    if (blk >= n_blocks) return;
    const uint8_t *base = src + src_off[blk];
    uint8_t *d = dst + dst_off[blk];
    const int64_t slot = lanes_slot_bytes(blk_off[blk + 1] - blk_off[blk], bps);
    __shared__ uint32_t start[65];
    if (tid == 0) {
        uint32_t acc = LANES_HDR;
        for (int l = 0; l < 64; ++l) { start[l] = acc; acc += ((const uint16_t *)base)[l]; }
        start[64] = acc;
    }
    for (int i = tid; i < LANES_HDR; i += 256) d[i] = base[i];
    __syncthreads();
    // 4 threads per lane stream, byte copies (streams start at arbitrary byte offsets of the packed file)
    const int l = tid >> 2, part = tid & 3;
    const uint32_t len = start[l + 1] - start[l];
    const uint8_t *sp = base + LANES_HDR + l * slot;
    uint8_t *dp = d + start[l];
    for (uint32_t i = part; i < len; i += 4) dp[i] = sp[i];
}

// Decoder, ONE WAVE PER BLOCK, one decoder per lane.  Symbol search per lane: a first guess from the inverse normal CDF
// of target / norm, then a walk on the exact integer CDF (the same gaussian_cdf_int as the encoder) with the division-free
// test cdf * span <= num.
// The 64 lane lengths of a block header, read bytewise (blocks and files sit back to back at arbitrary byte offsets in a
// container) and checked against the block's length from the header arrays: lane l's stream starts at *start; false (and
// *status = 1 + block) when the lengths do not add up — a truncated or corrupt container must not steer reads out of the blob.
__device__ __forceinline__ bool lanes_header(const uint8_t *base, int64_t block_bytes, int lane, int blk, uint32_t *start,
                                             int32_t *status) {
    const uint32_t mylen = (uint32_t)base[2 * lane] | ((uint32_t)base[2 * lane + 1] << 8);
    uint32_t incl = mylen;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    *start = incl - mylen;
    const uint32_t total = __shfl(incl, 63, 64);
    const bool ok = (int64_t)total + LANES_HDR == block_bytes;
    if (!ok && lane == 0 && status) atomicCAS(status, 0, 1 + blk);
    return ok;
}

__global__ void __launch_bounds__(64)
    gaussian_decode_lanes_kernel(const float *__restrict__ mean, const float *__restrict__ scale, const float *__restrict__ Q,
                                 int64_t q_div, const int64_t *__restrict__ blk_off, int n_blocks,
                                 const int32_t *__restrict__ min_v, const int32_t *__restrict__ max_v,
                                 const uint8_t *__restrict__ in, const int64_t *__restrict__ in_off, float *__restrict__ x_out,
                                 int32_t *__restrict__ status) {
    const int blk = blockIdx.x, lane = threadIdx.x;
    if (blk >= n_blocks) return;
    const int64_t b = blk_off[blk], e = blk_off[blk + 1];
    const uint8_t *base = in + in_off[blk];
    uint32_t start;
    if (!lanes_header(base, in_off[blk + 1] - in_off[blk], lane, blk, &start, status)) return;
    LaneAcDecoder dec;
    dec.init(base + LANES_HDR + start);
    const int lo = min_v[blk];
    const int Lp = max_v[blk] - lo + 2;
    if (Lp < 2 || Lp > 65536) {          // (the CDF has Lp + 1 <= 2^16 + 1 entries: a header outside that is corrupt)
        if (lane == 0 && status) atomicCAS(status, 0, 1 + blk);
        return;
    }
    const int max_sym = Lp - 2;
    const float norm = (float)(65536 - (Lp - 1));
    const float rnorm = 1.f / norm;
    // operands of step t + 1 are fetched before step t's search and consume: with one or two waves per SIMD nothing else
    // hides the load latency, and a lane's steps are a serial chain
    // (LANES_AHEAD steps: a step is ~2-4 us of serial work, an HBM miss up to ~2 us — one step ahead still left waits)
    float q_n[LANES_AHEAD], m_n[LANES_AHEAD], sc_n[LANES_AHEAD];
    LaneQIndex qix;
    qix.init(b + lane, q_div);
#pragma unroll
    for (int a = 0; a < LANES_AHEAD; ++a) {
        const int64_t j = b + lane + 64 * a;
        q_n[a] = 1.f; m_n[a] = 0.f; sc_n[a] = 1.f;
        if (j < e) { q_n[a] = Q[qix.quot]; m_n[a] = mean[j]; sc_n[a] = scale[j]; }
        qix.step();
    }
    for (int64_t i = b + lane; i < e; i += 64) {
        const float q = q_n[0], m = m_n[0], sc = sc_n[0];
#pragma unroll
        for (int a = 0; a + 1 < LANES_AHEAD; ++a) { q_n[a] = q_n[a + 1]; m_n[a] = m_n[a + 1]; sc_n[a] = sc_n[a + 1]; }
        {
            const int64_t j = i + 64 * LANES_AHEAD;
            if (j < e) { q_n[LANES_AHEAD - 1] = Q[qix.quot]; m_n[LANES_AHEAD - 1] = mean[j]; sc_n[LANES_AHEAD - 1] = scale[j]; }
            qix.step();
        }
        const float inv = 1.f / sc;
        const uint64_t num = dec.num();
        const uint32_t sm1 = dec.span_m1();
        // guess: target ~ num / span in [0, 2^16); cdf(j) ~ Phi(z_j) * norm + j
        // (v_rcp_f32 reciprocals: everything up to `guess` only steers the search, no decoded value depends on its rounding)
        const float tgt = (float)num * __builtin_amdgcn_rcpf((float)sm1 + 1.f);
        const float rq = __builtin_amdgcn_rcpf(q);
        // cdf(j) = round(Phi_j * norm) + j.  Left of ~-4.3 sigma the first term is 0 and the CDF is the ramp cdf(j) = j: the symbol
        // IS the target; right of +4.3 sigma it is the ramp norm + j: the symbol is target - norm.  In between, solve
        // Phi_j * norm + j = t for j with the inverse normal CDF (the + j term taken at the mean's index).  A wrong guess only
        // costs gallop steps; without the ramps a symbol far outside its predicted Gaussian (sigma at the 1e-9 clamp: every
        // symbol but one) cost ~2 log2(distance) erff evaluations and the whole wave waited for it.
        const float jm = m * rq - (float)lo;                                  // the mean, in symbol-index units
        const float hw = 4.3f * sc * rq;
        int guess;
        if (tgt <= jm - hw) guess = (int)tgt;
        else if (tgt - norm >= jm + hw + 1.f) guess = (int)(tgt - norm);
        else {
            const float u = fminf(fmaxf((tgt + 0.5f - jm) * rnorm, 1e-6f), 1.f - 1e-6f);
            const float z = SQRT2F * erfinvf(2.f * u - 1.f);
            guess = (int)floorf(jm + z * sc * rq + 0.5f);
        }
        guess = max(0, min(guess, max_sym));
        // largest sym in [0, max_sym] with cdf(sym) <= target.  First a WINDOW of four consecutive candidates around the guess,
        // evaluated unconditionally: four independent erff chains that the lane's instruction stream can overlap, no
        // data-dependent branch — and nearly always enough (the guess is the symbol or a neighbour).  Only a target outside
        // the window gallops away from its end and bisects (O(log distance) evaluations, never a linear walk).
        int sym;
        uint32_t c_low, c_high;
        {
            const int w0 = max(0, min(guess - 1, max_sym - 3));           // window [w0, w0 + 3] inside [0, max_sym] (or beyond, see ok)
            uint32_t cw[4];
            bool le[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cw[k] = gaussian_cdf_int(w0 + k, lo, norm, m, inv, q);
                le[k] = (w0 + k <= max_sym) && cdf_le_target(cw[k], sm1, num);
            }
            const bool inside = (le[0] || w0 == 0) && (!le[3] || w0 + 3 >= max_sym);
            if (inside) {
                const int k = (int)le[1] + (int)le[2] + (int)le[3];        // le is a prefix of trues: index of the last one (0 if none)
                sym = w0 + k;
                c_low = k == 0 ? cw[0] : (k == 1 ? cw[1] : (k == 2 ? cw[2] : cw[3]));
                c_high = sym >= max_sym ? AC_TOP : (k == 0 ? cw[1] : (k == 1 ? cw[2] : (k == 2 ? cw[3]
                                                                   : gaussian_cdf_int(sym + 1, lo, norm, m, inv, q))));
            } else {
                int lo_j, hi_j;                     // le(lo_j) holds (or lo_j = 0), le(hi_j) fails (or hi_j = max_sym + 1)
                uint32_t c_lo = 0, c_hi = AC_TOP;
                bool lo_known = false;
                if (le[3]) {                        // the target lies above the window
                    lo_j = w0 + 3; c_lo = cw[3]; lo_known = true;
                    int d = 4;
                    hi_j = max_sym + 1;
                    while (lo_j + d <= max_sym) {
                        const uint32_t cc = gaussian_cdf_int(lo_j + d, lo, norm, m, inv, q);
                        if (cdf_le_target(cc, sm1, num)) { lo_j += d; c_lo = cc; d <<= 1; }
                        else { hi_j = lo_j + d; c_hi = cc; break; }
                    }
                } else {                            // below it
                    hi_j = w0; c_hi = cw[0];
                    int d = 4;
                    lo_j = 0;
                    while (hi_j - d > 0) {
                        const uint32_t cc = gaussian_cdf_int(hi_j - d, lo, norm, m, inv, q);
                        if (!cdf_le_target(cc, sm1, num)) { hi_j -= d; c_hi = cc; d <<= 1; }
                        else { lo_j = hi_j - d; c_lo = cc; lo_known = true; break; }
                    }
                }
                while (hi_j - lo_j > 1) {
                    const int mid = (lo_j + hi_j) >> 1;
                    const uint32_t cc = gaussian_cdf_int(mid, lo, norm, m, inv, q);
                    if (cdf_le_target(cc, sm1, num)) { lo_j = mid; c_lo = cc; lo_known = true; }
                    else { hi_j = mid; c_hi = cc; }
                }
                sym = lo_j;
                c_low = lo_known ? c_lo : gaussian_cdf_int(sym, lo, norm, m, inv, q);
                c_high = sym >= max_sym ? AC_TOP : c_hi;
            }
        }
        x_out[i] = (float)(sym + lo) * q;
        if (i + 64 < e) dec.consume(c_low, c_high);
    }
}

extern "C" int cgs_gaussian_ac_encode_lanes(const float *x, const float *mean, const float *scale, const float *Q,
                                            int64_t q_div, const int64_t *blk_off, int n_blocks, const int32_t *min_v,
                                            const int32_t *max_v, uint8_t *out, const int64_t *out_off, uint32_t *out_len,
                                            int32_t *status, void *stream) {
    if (n_blocks < 0 || q_div < 1) { cgs_set_error("gaussian_ac_encode_lanes: bad args"); return CGS_ERR_ARG; }
    if (n_blocks == 0) return CGS_OK;
    hipLaunchKernelGGL(gaussian_encode_lanes_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, x, mean, scale, Q,
                       q_div, blk_off, n_blocks, min_v, max_v, out, out_off, out_len, status);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_lanes_compact(const uint8_t *src, const int64_t *src_off, const int64_t *blk_off, const int64_t *dst_off,
                                 int n_blocks, uint8_t *dst, int bytes_per_symbol, void *stream) {
    if (n_blocks < 0 || bytes_per_symbol < 1) { cgs_set_error("lanes_compact: bad args"); return CGS_ERR_ARG; }
    if (n_blocks == 0) return CGS_OK;
    hipLaunchKernelGGL(lanes_compact_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, src, src_off, blk_off, dst_off,
                       n_blocks, dst, bytes_per_symbol);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_gaussian_ac_decode_lanes(const float *mean, const float *scale, const float *Q, int64_t q_div,
                                            const int64_t *blk_off, int n_blocks, const int32_t *min_v, const int32_t *max_v,
                                            const uint8_t *in, const int64_t *in_off, float *x_out, int32_t *status,
                                            void *stream) {
    if (n_blocks < 0 || q_div < 1) { cgs_set_error("gaussian_ac_decode_lanes: bad args"); return CGS_ERR_ARG; }
    if (n_blocks == 0) return CGS_OK;
    hipLaunchKernelGGL(gaussian_decode_lanes_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, mean, scale, Q, q_div,
                       blk_off, n_blocks, min_v, max_v, in, in_off, x_out, status);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// ---------------------------------------------------------------------------------
// Lane-parallel table codec (container version 2: hyper.b)
// ---------------------------------------------------------------------------------
// The hyper latents (scene/gaussian_model.py:1082-1098,1326-1338: compressai's EntropyBottleneck.compress, a per-channel
// frequency table, rANS strings of 10 000 anchors; the wheel is not in the mount) in version 2: the SAME per-channel integer
// tables (EntropyBottleneck.update), the arithmetic coder of this file, lane-parallel blocks as above.  A block holds
// consecutive anchors of ONE channel (blk_ch), so its table is staged once.  A value outside the table's support is the
// escape slot followed by its sign, the bit length of m = distance + 1 in unary and m's low bits, each as an equiprobable
// binary symbol (the same payload the host rANS coder sends).
#define TAB_MAX_LEN 1024
#define TAB_BPS 6            // worst-case bytes per symbol of a lane slot (escapes cost up to ~10 bytes; an overflow is reported)

template <class Enc>
__device__ __forceinline__ void table_put_bit(Enc &enc, uint32_t bit) { enc.encode(bit ? 0x8000u : 0u, bit ? AC_TOP : 0x8000u); }

__global__ void __launch_bounds__(64)
    table_encode_lanes_kernel(const int32_t *__restrict__ sym, const int64_t *__restrict__ blk_off, const int32_t *__restrict__ blk_ch,
                              int n_blocks, const int32_t *__restrict__ cdf, int max_len, const int32_t *__restrict__ cdf_len,
                              const int32_t *__restrict__ offset, uint8_t *__restrict__ out, const int64_t *__restrict__ out_off,
                              uint32_t *__restrict__ out_len, int32_t *__restrict__ status) {
    __shared__ uint32_t tab[TAB_MAX_LEN];
    const int blk = blockIdx.x, lane = threadIdx.x;
    if (blk >= n_blocks) return;
    const int ch = blk_ch[blk];
    const int len = cdf_len[ch], max_value = len - 2, off = offset[ch];
    for (int i = lane; i < len; i += 64) tab[i] = (uint32_t)cdf[(size_t)ch * max_len + i];
    __syncthreads();
    const int64_t b = blk_off[blk], e = blk_off[blk + 1];
    const int64_t slot = lanes_slot_bytes(e - b, TAB_BPS);
    uint8_t *base = out + out_off[blk];
    uint8_t *mine = base + LANES_HDR + lane * slot;
    AcEncoderT<WaveBitWriter> enc;
    enc.init(mine, (size_t)slot);
    for (int64_t i = b + lane; i < e; i += 64) {
        const int raw = sym[i] - off;
        const bool esc = raw < 0 || raw >= max_value;
        const int v = esc ? max_value : raw;
        enc.encode(tab[v], tab[v + 1]);
        if (esc) {
            const uint32_t m = raw < 0 ? (uint32_t)(-raw) : (uint32_t)(raw - max_value + 1);
            table_put_bit(enc, raw < 0 ? 1u : 0u);
            const int nb = 31 - __clz((int)m);                      // floor(log2 m), m >= 1
            for (int k = 0; k < nb; ++k) table_put_bit(enc, 0u);
            table_put_bit(enc, 1u);
            for (int k = nb - 1; k >= 0; --k) table_put_bit(enc, (m >> k) & 1u);
        }
    }
    const uint32_t bytes = (uint32_t)enc.finish(mine);
    ((uint16_t *)base)[lane] = (uint16_t)bytes;
    uint32_t tot = bytes;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d, 64);
    if (__ballot(enc.out.overflow || bytes > 65535u) != 0ull && lane == 0) atomicMax(status, 2);
    if (lane == 0) out_len[blk] = LANES_HDR + tot;
}

// -> dequantised rows: out_rows[(i - ch * n_per_ch) * ld + ch] = (float)symbol + medians[ch]
__global__ void __launch_bounds__(64)
    table_decode_lanes_kernel(const int64_t *__restrict__ blk_off, const int32_t *__restrict__ blk_ch, int n_blocks,
                              const int32_t *__restrict__ cdf, int max_len, const int32_t *__restrict__ cdf_len,
                              const int32_t *__restrict__ offset, const float *__restrict__ medians, int64_t n_per_ch,
                              const uint8_t *__restrict__ in, const int64_t *__restrict__ in_off, float *__restrict__ out_rows,
                              int64_t ld, int32_t *__restrict__ status) {
    __shared__ uint32_t tab[TAB_MAX_LEN];
    const int blk = blockIdx.x, lane = threadIdx.x;
    if (blk >= n_blocks) return;
    const int ch = blk_ch[blk];
    const int len = cdf_len[ch], max_value = len - 2, off = offset[ch];
    for (int i = lane; i < len; i += 64) tab[i] = (uint32_t)cdf[(size_t)ch * max_len + i];
    __syncthreads();
    const float med = medians[ch];
    const int64_t b = blk_off[blk], e = blk_off[blk + 1];
    const uint8_t *base = in + in_off[blk];
    uint32_t start;
    if (!lanes_header(base, in_off[blk + 1] - in_off[blk], lane, blk, &start, status)) return;
    LaneAcDecoder dec;
    dec.init(base + LANES_HDR + start);
    auto get_bit = [&]() -> uint32_t {
        const uint32_t bit = cdf_le_target(0x8000u, dec.span_m1(), dec.num()) ? 1u : 0u;
        dec.consume(bit ? 0x8000u : 0u, bit ? AC_TOP : 0x8000u);
        return bit;
    };
    for (int64_t i = b + lane; i < e; i += 64) {
        const uint64_t num = dec.num();
        const uint32_t sm1 = dec.span_m1();
        int lo_j = 0, hi_j = max_value + 1;          // largest v in [0, max_value] with tab[v] <= target (tab[0] = 0)
        while (hi_j - lo_j > 1) {
            const int mid = (lo_j + hi_j) >> 1;
            if (cdf_le_target(tab[mid], sm1, num)) lo_j = mid; else hi_j = mid;
        }
        int value = lo_j;
        dec.consume(tab[value], tab[value + 1]);     // (every symbol is consumed: escape payloads may follow the last one)
        if (value == max_value) {
            const uint32_t sign = get_bit();
            int nz = 0;
            while (get_bit() == 0u && nz < 31) ++nz;
            uint32_t m = 1;
            for (int k = 0; k < nz; ++k) m = (m << 1) | get_bit();
            value = sign ? -(int)m : (int)m + max_value - 1;
        }
        out_rows[(i - (int64_t)ch * n_per_ch) * ld + ch] = (float)(value + off) + med;
    }
}

extern "C" int cgs_table_ac_encode_lanes(const int32_t *sym, const int64_t *blk_off, const int32_t *blk_ch, int n_blocks,
                                         const int32_t *cdf, int max_len, const int32_t *cdf_len, const int32_t *offset,
                                         uint8_t *out, const int64_t *out_off, uint32_t *out_len, int32_t *status, void *stream) {
    if (n_blocks < 0 || max_len < 3 || max_len > TAB_MAX_LEN) { cgs_set_error("table_ac_encode_lanes: bad args (tables of up to %d entries)", TAB_MAX_LEN); return CGS_ERR_ARG; }
    if (n_blocks == 0) return CGS_OK;
    hipLaunchKernelGGL(table_encode_lanes_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, sym, blk_off, blk_ch, n_blocks,
                       cdf, max_len, cdf_len, offset, out, out_off, out_len, status);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_table_ac_decode_lanes(const int64_t *blk_off, const int32_t *blk_ch, int n_blocks, const int32_t *cdf,
                                         int max_len, const int32_t *cdf_len, const int32_t *offset, const float *medians,
                                         int64_t n_per_channel, const uint8_t *in, const int64_t *in_off, float *out_rows,
                                         int64_t ld_rows, int32_t *status, void *stream) {
    if (n_blocks < 0 || max_len < 3 || max_len > TAB_MAX_LEN) { cgs_set_error("table_ac_decode_lanes: bad args"); return CGS_ERR_ARG; }
    if (n_blocks == 0) return CGS_OK;
    hipLaunchKernelGGL(table_decode_lanes_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, blk_off, blk_ch, n_blocks, cdf,
                       max_len, cdf_len, offset, medians, n_per_channel, in, in_off, out_rows, ld_rows, status);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// ---------------------------------------------------------------------------------
// range-ANS for the hyper-prior symbols (host)
// ---------------------------------------------------------------------------------
// 32-bit state, 16-bit renormalisation words, frequency precision `prec` (16).
// Symbols are coded in REVERSE so that the decoder reads them forward.  Channel c's
// table cdf[c] has cdf_len[c] entries; index value = symbol - offset[c]; the last
// table slot (cdf_len-2) is the escape: out-of-support values are sent as escape
// followed by (sign, Elias-gamma(magnitude)) in 1-bit uniform rANS steps.
#define RANS_L (1u << 16)

struct RansEnc {
    uint32_t x;
    std::vector<uint16_t> words;   // emitted in reverse order
    RansEnc() : x(RANS_L) {}
    void put(uint32_t start, uint32_t freq, int prec) {
        const uint32_t x_max = ((RANS_L >> prec) << 16) * freq;
        while (x >= x_max) { words.push_back((uint16_t)(x & 0xFFFFu)); x >>= 16; }
        x = ((x / freq) << prec) + (x % freq) + start;
    }
    void put_bits(uint32_t v, int nbits) { for (int i = 0; i < nbits; ++i) put((v >> i) & 1u, 1, 1); }   // LSB first pushed => MSB first popped? see decoder
};

struct RansDec {
    uint32_t x;
    const uint16_t *p, *end;
    void init(const uint8_t *buf, size_t len) {
        p = (const uint16_t *)(buf + 4);
        end = (const uint16_t *)(buf + (len & ~(size_t)1));
        x = (uint32_t)buf[0] | ((uint32_t)buf[1] << 8) | ((uint32_t)buf[2] << 16) | ((uint32_t)buf[3] << 24);
    }
    uint32_t peek(int prec) const { return x & ((1u << prec) - 1u); }
    void advance(uint32_t start, uint32_t freq, int prec) {
        x = freq * (x >> prec) + (x & ((1u << prec) - 1u)) - start;
        while (x < RANS_L && p < end) { x = (x << 16) | *p++; }
    }
    uint32_t get_bit() { const uint32_t b = peek(1); advance(b, 1, 1); return b; }
};

// Escape payload: sign bit, then n = bit-length of m (unary: n-1 zeros then a one), then the n-1 low bits of m,
// where m = out-of-support distance + 1 >= 1.
static void esc_bits(int value, int max_value, std::vector<int> &bits) {
    uint32_t m;
    int sign;
    if (value < 0) { sign = 1; m = (uint32_t)(-value); } else { sign = 0; m = (uint32_t)(value - max_value + 1); }
    bits.push_back(sign);
    int n = 0;
    while ((m >> n) > 1) ++n;            // n = floor(log2 m)
    for (int i = 0; i < n; ++i) bits.push_back(0);
    bits.push_back(1);
    for (int i = n - 1; i >= 0; --i) bits.push_back((m >> i) & 1);
}

extern "C" size_t cgs_rans_max_bytes(int64_t n_sym) { return (size_t)(n_sym > 0 ? n_sym : 0) * 12 + 64; }

// symbols int32 [C, n] (channel-major), cdf int32 [C, max_len], cdf_len [C], offset [C]
extern "C" int cgs_rans_encode_host(const int32_t *symbols, int C_, int64_t n, const int32_t *cdf, int max_len,
                                    const int32_t *cdf_len, const int32_t *offset, int prec, uint8_t *out,
                                    size_t out_cap, size_t *out_len) {
    if (!symbols || !cdf || !cdf_len || !offset || !out || !out_len || prec < 8 || prec > 16) {
        cgs_set_error("rans_encode: bad args");
        return CGS_ERR_ARG;
    }
    RansEnc enc;
    std::vector<int> bits;
    // coding order (decoder's view): for i in 0..n-1, for c in 0..C-1  -> encode in reverse
    for (int64_t i = n - 1; i >= 0; --i)
        for (int c = C_ - 1; c >= 0; --c) {
            const int32_t *tab = cdf + (size_t)c * max_len;
            const int max_value = cdf_len[c] - 2;          // escape slot
            int value = symbols[(size_t)c * n + i] - offset[c];
            bool esc = false;
            int raw = value;
            if (value < 0 || value >= max_value) { esc = true; value = max_value; }
            if (esc) {
                bits.clear();
                esc_bits(raw, max_value, bits);
                for (int k = (int)bits.size() - 1; k >= 0; --k) enc.put((uint32_t)bits[k], 1, 1);
            }
            enc.put((uint32_t)tab[value], (uint32_t)(tab[value + 1] - tab[value]), prec);
        }
    const size_t need = 4 + enc.words.size() * 2;
    if (need > out_cap) { cgs_set_error("rans_encode: output buffer too small"); return CGS_ERR_WORKSPACE; }
    out[0] = (uint8_t)enc.x; out[1] = (uint8_t)(enc.x >> 8); out[2] = (uint8_t)(enc.x >> 16); out[3] = (uint8_t)(enc.x >> 24);
    uint8_t *p = out + 4;
    for (size_t k = enc.words.size(); k-- > 0;) { *p++ = (uint8_t)enc.words[k]; *p++ = (uint8_t)(enc.words[k] >> 8); }
    *out_len = need;
    return CGS_OK;
}

// One hyper-prior chunk string -> symbols.  Two output forms: int32 symbols[C, n] (channel-major, the layout
// compressai's decompress returns), or — out_rows != NULL — the DEQUANTISED latents, row-major by anchor:
// out_rows[i * ld_rows + c] = (float)symbol + medians[c], which is what the container decoder feeds the context model
// (scene/gaussian_model.py:1326-1338): every chunk job writes its rows of one pinned [N, C] buffer, no concatenation,
// transpose or integer-to-float pass afterwards.
static int rans_decode_impl(const uint8_t *in, size_t in_len, int C_, int64_t n, const int32_t *cdf, int max_len,
                            const int32_t *cdf_len, const int32_t *offset, int prec, int32_t *symbols,
                            const float *medians, float *out_rows, int64_t ld_rows) {
    RansDec dec;
    dec.init(in, in_len);
    // Symbol search: the largest v in [0, max_value] with tab[v] <= t.  A 256-slot table per channel over the top 8 bits of
    // t gives the answer for the slot's first value, a short forward scan the rest (0-1 steps for the ~60-entry tables of
    // the hyper prior) — a binary search is six unpredictable branches per symbol, and the 12 M hyper symbols of a 1 M-anchor
    // container sit on the decoder's critical path.  Same symbols (the scan stops at exactly the searched v).
    const bool use_lut = prec >= 8 && prec <= 24 && C_ <= 64 && n * C_ >= 4096;
    uint16_t lut[64 * 256];
    if (use_lut)
        for (int c = 0; c < C_; ++c) {
            const int32_t *tab = cdf + (size_t)c * max_len;
            const int max_value = cdf_len[c] - 2;
            int v = 0;
            for (int sl = 0; sl < 256; ++sl) {
                const uint32_t t0 = (uint32_t)sl << (prec - 8);
                while (v < max_value && (uint32_t)tab[v + 1] <= t0) ++v;
                lut[c * 256 + sl] = (uint16_t)v;
            }
        }
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < C_; ++c) {
            const int32_t *tab = cdf + (size_t)c * max_len;
            const int max_value = cdf_len[c] - 2;
            const uint32_t t = dec.peek(prec);
            int value;
            if (use_lut) {
                value = lut[c * 256 + (t >> (prec - 8))];
                while (value < max_value && (uint32_t)tab[value + 1] <= t) ++value;
            } else {
                int lo = 0, hi = max_value + 1;            // largest v with tab[v] <= t
                while (lo + 1 < hi) { const int m = (lo + hi) >> 1; if ((uint32_t)tab[m] <= t) lo = m; else hi = m; }
                value = lo;
            }
            dec.advance((uint32_t)tab[value], (uint32_t)(tab[value + 1] - tab[value]), prec);
            if (value == max_value) {
                const int sign = (int)dec.get_bit();
                int nz = 0;
                while (dec.get_bit() == 0) { if (++nz > 31) { cgs_set_error("rans_decode: corrupt escape"); return CGS_ERR_BOUNDS; } }
                uint32_t m = 1;
                for (int k = 0; k < nz; ++k) m = (m << 1) | dec.get_bit();
                value = sign ? -(int)m : (int)m + max_value - 1;
            }
            if (out_rows) out_rows[(size_t)i * ld_rows + c] = (float)(value + offset[c]) + medians[c];
            else symbols[(size_t)c * n + i] = value + offset[c];
        }
    return CGS_OK;
}

extern "C" int cgs_rans_decode_host(const uint8_t *in, size_t in_len, int C_, int64_t n, const int32_t *cdf, int max_len,
                                    const int32_t *cdf_len, const int32_t *offset, int prec, int32_t *symbols) {
    if (!in || in_len < 4 || !cdf || !cdf_len || !offset || !symbols) { cgs_set_error("rans_decode: bad args"); return CGS_ERR_ARG; }
    return rans_decode_impl(in, in_len, C_, n, cdf, max_len, cdf_len, offset, prec, symbols, nullptr, nullptr, 0);
}

extern "C" int cgs_rans_decode_rows_host(const uint8_t *in, size_t in_len, int C_, int64_t n, const int32_t *cdf,
                                         int max_len, const int32_t *cdf_len, const int32_t *offset, int prec,
                                         const float *medians, float *out_rows, int64_t ld_rows) {
    if (!in || in_len < 4 || !cdf || !cdf_len || !offset || !medians || !out_rows || ld_rows < C_) {
        cgs_set_error("rans_decode_rows: bad args");
        return CGS_ERR_ARG;
    }
    return rans_decode_impl(in, in_len, C_, n, cdf, max_len, cdf_len, offset, prec, nullptr, medians, out_rows, ld_rows);
}
