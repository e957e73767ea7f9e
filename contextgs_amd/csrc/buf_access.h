// Branch-free global access through RAW BUFFER instructions (hardware bounds check), shared by the fused level kernels
// (ctx_level.hip) and the fused anchor-MLP kernels (mlp3.hip).
//
// A lane that has nothing to load — a row past the end, a piece another lane group owns — issues the same instruction with an
// out-of-range offset and gets zeros: NO branch around the load, so no exec-mask join at which the compiler has to wait for it
// (predicated global loads cost a full `s_waitcnt vmcnt(0)` at every join, which serialises a prefetch into its pieces:
// profiles/r05_ctx_level.txt).  Offsets are 32-bit: an operand must stay below CL_MAX_BYTES (the host checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mlp_frag.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x3 __attribute__((ext_vector_type(3)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t ClBuf;
#ifndef CL_ST_AUX
#define CL_ST_AUX 0               // cache policy of the buffer stores (2 = non-temporal)
#endif
#ifndef CL_LD_AUX
#define CL_LD_AUX 0               // cache policy of the buffer loads
#endif
#define CL_OOB 0xFFFFFF00u              // an offset no buffer reaches (the host refuses buffers >= 0xFFFFF000 bytes)
#define CL_MAX_BYTES 0xFFFFF000ull

__device__ __forceinline__ ClBuf cl_buf(const void *p, uint64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, p ? (int)(uint32_t)bytes : 0, 0x00020000);
}
__device__ __forceinline__ f32x4 cl_l128(ClBuf b, uint32_t off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b, (int)off, 0, CL_LD_AUX));
}
// (results and operands cross between int and float vectors by WHOLE-vector bit casts only: extracting .x / .y / .z from the
//  builtins' int vectors came back as the first component replicated — tools/micro/buf_probe.hip)
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 cl_l96(ClBuf b, uint32_t off) {
    const f32x3 v = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(b, (int)off, 0, CL_LD_AUX));
    return (f32x4){v[0], v[1], v[2], 0.f};
}
__device__ __forceinline__ f32x4 cl_l64(ClBuf b, uint32_t off) {
    const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(b, (int)off, 0, CL_LD_AUX));
    return (f32x4){v[0], v[1], 0.f, 0.f};
}
__device__ __forceinline__ float cl_l32(ClBuf b, uint32_t off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b, (int)off, 0, CL_LD_AUX));
}
__device__ __forceinline__ int64_t cl_li64(ClBuf b, uint32_t off) {
    return __builtin_bit_cast(int64_t, __builtin_amdgcn_raw_buffer_load_b64(b, (int)off, 0, CL_LD_AUX));
}
__device__ __forceinline__ void cl_s128(ClBuf b, uint32_t off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), b, (int)off, 0, CL_ST_AUX);
}
__device__ __forceinline__ void cl_s96(ClBuf b, uint32_t off, f32x4 v) {
    const f32x3 t = (f32x3){v[0], v[1], v[2]};
    __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(i32x3, t), b, (int)off, 0, CL_ST_AUX);
}
__device__ __forceinline__ void cl_s64(ClBuf b, uint32_t off, f32x4 v) {
    const f32x2 t = (f32x2){v[0], v[1]};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, t), b, (int)off, 0, CL_ST_AUX);
}
__device__ __forceinline__ void cl_s32(ClBuf b, uint32_t off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), b, (int)off, 0, CL_ST_AUX);
}
__device__ __forceinline__ uint32_t cl_sel(bool on, uint32_t off) { return on ? off : CL_OOB; }


// columns [16q + 4g, +4) of a row of DIM floats at byte offset `rowoff` of buffer b (the fragment layout of mlp_frag.h); a piece
// that starts inside the row but ends behind it carries the next row's first values: frag_bmask4 zeroes them at use
template <int DIM>
__device__ __forceinline__ f32x4 frag_bload4(ClBuf b, uint32_t rowoff, int q, int g, bool on) {
    const int col0 = 16 * q + 4 * g;
    return cl_l128(b, cl_sel(on && col0 < DIM, rowoff + (uint32_t)col0 * 4));
}
template <int DIM>
__device__ __forceinline__ f32x4 frag_bmask4(f32x4 v, int q, int g) {
    if (16 * q + 15 >= DIM) {          // only the tile that holds the row end
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (16 * q + 4 * g + j >= DIM) v[j] = 0.f;
    }
    return v;
}
// store of such a piece: 16 bytes when it lies inside the row, 12 / 8 / 4 when the row ends inside it
template <int DIM>
__device__ __forceinline__ void frag_bstore4(ClBuf b, uint32_t rowoff, int q, int g, bool on, f32x4 v) {
    const int col0 = 16 * q + 4 * g;
    const uint32_t o = rowoff + (uint32_t)col0 * 4;
    if (16 * q + 15 < DIM) { cl_s128(b, cl_sel(on, o), v); return; }
    cl_s128(b, cl_sel(on && col0 + 3 < DIM, o), v);
    if (DIM % 4 == 3) cl_s96(b, cl_sel(on && col0 + 3 == DIM, o), v);
    if (DIM % 4 == 2) cl_s64(b, cl_sel(on && col0 + 2 == DIM, o), v);
    if (DIM % 4 == 1) cl_s32(b, cl_sel(on && col0 + 1 == DIM, o), v[0]);
}

// The same pieces of a hand-over buffer in FRAGMENT-MAJOR ("tiled") form: per 16-row tile (DIM * 64 bytes at `tilebase`) the full
// 16-column pieces come first, 1 KB each in LANE order (lane 16 g + c -> its 16 bytes: one store / load instruction of a wave is one
// contiguous, aligned KB instead of sixteen 64-byte chunks DIM * 4 bytes apart that straddle sectors), then the DIM % 16 tail
// columns row by row.  c = the lane's row of the tile.  Same bytes as the row-major form (rows padded to a multiple of 16).
template <int DIM>
__device__ __forceinline__ uint32_t frag_toff(int q, int g, int c) {
    constexpr int FULL = DIM / 16, TAIL = DIM % 16;
    return q < FULL ? (uint32_t)(q * 1024 + (16 * g + c) * 16) : (uint32_t)(FULL * 1024 + c * (TAIL * 4) + 16 * g);
}
template <int DIM>
__device__ __forceinline__ f32x4 frag_tload4(ClBuf b, uint32_t tilebase, int q, int g, int c, bool on) {
    const int col0 = 16 * q + 4 * g;
    return cl_l128(b, cl_sel(on && col0 < DIM, tilebase + frag_toff<DIM>(q, g, c)));
}
template <int DIM>
__device__ __forceinline__ void frag_tstore4(ClBuf b, uint32_t tilebase, int q, int g, int c, bool on, f32x4 v) {
    const int col0 = 16 * q + 4 * g;
    const uint32_t o = tilebase + frag_toff<DIM>(q, g, c);
    if (16 * q + 15 < DIM) { cl_s128(b, cl_sel(on, o), v); return; }
    cl_s128(b, cl_sel(on && col0 + 3 < DIM, o), v);
    if (DIM % 4 == 3) cl_s96(b, cl_sel(on && col0 + 3 == DIM, o), v);
    if (DIM % 4 == 2) cl_s64(b, cl_sel(on && col0 + 2 == DIM, o), v);
    if (DIM % 4 == 1) cl_s32(b, cl_sel(on && col0 + 1 == DIM, o), v[0]);
}

// A 16-row tile of a ROW-MAJOR [*, DIM] array stored from its fragments through a per-wave LDS patch of 16 * DIM floats: the
// fragments are written to the patch in row-major order, and the patch — byte for byte the tile's memory image, 64 * DIM contiguous
// bytes at `tile_off` of b — leaves as contiguous 1 KB store instructions instead of sixteen 64-byte chunks a row apart per
// fragment.  Rows behind the end of b are dropped by its bounds check (the tile's valid rows are a prefix of its image).
// LDS operations of one wave execute in order: no barrier between the two phases, nor before the patch's next use.
typedef float f32x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
template <int DIM, int NT>
__device__ __forceinline__ void frag_tile_store(float *patch, ClBuf b, uint32_t tile_off, const f32x4 (&v)[NT], int g, int c, int lane) {
    static_assert(DIM % 2 == 0, "8-byte aligned pieces");
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int col0 = 16 * u + 4 * g;
        float *p = patch + c * DIM + col0;
        const f32x2_a8 lo = (f32x2_a8){v[u][0], v[u][1]}, hi = (f32x2_a8){v[u][2], v[u][3]};
        if (16 * u + 15 < DIM) {
            *(f32x2_a8 *)p = lo;
            *(f32x2_a8 *)(p + 2) = hi;
        } else {
            if (col0 + 1 < DIM) *(f32x2_a8 *)p = lo;
            if (col0 + 3 < DIM) *(f32x2_a8 *)(p + 2) = hi;
        }
    }
    constexpr int TB = 64 * DIM;
#pragma unroll
    for (int k = 0; k < (TB + 1023) / 1024; ++k) {
        const int off = k * 1024 + lane * 16;
        const bool in = (k + 1) * 1024 <= TB || off < TB;
        const f32x4 t = *(const f32x4 *)((const char *)patch + (in ? off : 0));
        cl_s128(b, cl_sel(in, tile_off + (uint32_t)off), t);
    }
}

// The two halves of frag_tile_store for tiles assembled from pieces of other shapes (ctx_level.hip): cl_patch_put4 writes the first
// `cnt` values of a piece at a 4-byte aligned LDS address, cl_patch_flush stores TB bytes of the patch (16-byte aligned, TB a
// multiple of 16) to byte offset `off` of b as contiguous 16-byte-per-lane stores.
__device__ __forceinline__ void cl_patch_put4(float *p, f32x4 v, int cnt) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < cnt) p[j] = v[j];
}
template <int TB>
__device__ __forceinline__ void cl_patch_flush(const float *patch, ClBuf b, uint32_t off, int lane) {
    static_assert(TB % 16 == 0, "whole 16-byte pieces");
#pragma unroll
    for (int k = 0; k < (TB + 1023) / 1024; ++k) {
        const int o = k * 1024 + lane * 16;
        const bool in = (k + 1) * 1024 <= TB || o < TB;
        const f32x4 t = *(const f32x4 *)((const char *)patch + (in ? o : 0));
        cl_s128(b, cl_sel(in, off + (uint32_t)o), t);
    }
}
