// Gemv: out[:,y,z] = m[:,:,z] * v[:,y,z]  or  m[:,:,z]^T * v[:,y,z]      (wgebra gemv.wgsl:28-155)
//
// HBM-bound: the matrix (4*R*C bytes) is streamed exactly once; v / out are noise. Column-major storage decides
// the two shapes:
//
//  N  (out = m v):  a column is contiguous over rows, so lane l owns rows 4l..4l+3 (one float4) and a wave sweeps
//     columns: every load is 16 B/lane, 1 KiB contiguous per wave. A 256-thread workgroup covers 256 rows; its 4
//     waves take disjoint column ranges and are summed through LDS. Because R/256 row blocks cannot fill 256 CUs
//     (config 4: R = 4096 -> 16), the column range is additionally split over grid.y ("split-K"); each split
//     writes a partial and a tiny second kernel sums the partials in a FIXED order (deterministic; no atomics).
//     v is fetched 64 columns at a time with one coalesced load per wave and broadcast with v_readlane.
//
//  T  (out = m^T v): a column is one dot product with v. Each wave owns 4 adjacent columns and sweeps rows with
//     float4 loads (v's float4 is loaded once and reused for the 4 columns), then a wave-shuffle butterfly
//     reduces the 64 lanes; the 4 results are stored as one float4. Rows are split over grid.y when there are too
//     few columns to fill the chip.
//
// The reference's four variants differ only in how invocations are laid out (naive: one thread per 4 rows; fast: a
// 32-lane workgroup + LDS tree per 4 rows) and in their summation order; all four map to these two kernels. GEMV
// parity is tolerance-based: the order here is "per-lane sequential, then butterfly", which satisfies the
// same error bound as both WGSL orders.
#include "wg_internal.hpp"
#include "reduce_ops.hpp"
#include <cstdlib>

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxRhs = 8; // RHS columns handled per pass over the matrix (m is read once for all of them)

#ifndef GEMV_NT
#define GEMV_NT 1
#endif
#ifndef GEMV_NU
#define GEMV_NU 4 // N kernel: columns (float4 loads) in flight per lane (4: 6.45 TB/s, 8/16: 6.40 at 4096 x 65536)
#endif
#ifndef WG_GEMV_TS_F16
#define WG_GEMV_TS_F16 2
#endif
#ifndef GEMV_TS
#define GEMV_TS 1 // T kernel: row-steps (4 float4 each) in flight per lane (1: 6.40 TB/s, 2/4: 6.34)
#endif
__device__ __forceinline__ float4 ld_stream(const float4 *p) { return GEMV_NT ? wg_ld_nt(p) : *p; }
__device__ __forceinline__ float readlane_f(float x, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}
__device__ __forceinline__ void fma4(float4 &acc, float4 a, float s) {
    acc.x = fmaf(a.x, s, acc.x);
    acc.y = fmaf(a.y, s, acc.y);
    acc.z = fmaf(a.z, s, acc.z);
    acc.w = fmaf(a.w, s, acc.w);
}
// Element types: f32 (the reference's), and f16 as this build's extension (f16 matrix / vectors, f32 accumulation in the same per-lane +
// butterfly order, one rounding when the result is stored; split partials stay f32). Four consecutive elements as floats:
typedef _Float16 wg_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load4s(const float *p) { return ld_stream(reinterpret_cast<const float4 *>(p)); } // streaming (matrix)
__device__ __forceinline__ float4 load4s(const _Float16 *p) {
    const wg_h4 v = GEMV_NT ? __builtin_nontemporal_load(reinterpret_cast<const wg_h4 *>(p)) : *reinterpret_cast<const wg_h4 *>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); } // cached (vector)
__device__ __forceinline__ float4 load4(const _Float16 *p) {
    const wg_h4 v = *reinterpret_cast<const wg_h4 *>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void store4(float *d, float4 s) { *reinterpret_cast<float4 *>(d) = s; }
__device__ __forceinline__ void store4(_Float16 *d, float4 s) {
    const wg_h4 v = { (_Float16)s.x, (_Float16)s.y, (_Float16)s.z, (_Float16)s.w };
    *reinterpret_cast<wg_h4 *>(d) = v;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) x += __shfl_xor(x, s, 64);
    return x;
}

template <typename T>
struct GemvArgsT {
    const T *m; uint32_t ldm; uint64_t m_batch;
    const T *v; uint32_t ldv; uint64_t v_batch;
    T *out;          // the result (written when part == nullptr) ...
    float *part;     // ... or the f32 partials buffer when nsplit > 1
    uint32_t ld_dst; // elements between RHS columns of the destination
    uint64_t dst_batch;
    uint64_t dst_split; // elements between splits (partials only)
    uint32_t rows_out;  // length of out
    uint32_t k;         // contraction length
    uint32_t nrhs;      // total RHS columns
    uint32_t k_per_split;
};
using GemvArgs = GemvArgsT<float>;

// ------------------------------------------------------------------------------------------------------
// N: dst[r] = sum_c m[r, c] v[c],  r in this block's 256 rows, c in this split's column range
// grid = (row blocks, splits, nmats * rhs groups)
// ------------------------------------------------------------------------------------------------------
template <int NRHS, typename T>
__global__ __launch_bounds__(kThreads) void gemv_n_kernel(GemvArgsT<T> a) {
    __shared__ float4 part[kWaves][NRHS][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t rhs_groups = (a.nrhs + kMaxRhs - 1) / kMaxRhs;
    const uint32_t z = blockIdx.z / rhs_groups, y0 = (blockIdx.z % rhs_groups) * kMaxRhs;
    const uint32_t row = blockIdx.x * 256u + 4u * lane;
    const bool row_ok = row < a.rows_out;

    const uint32_t c_begin = blockIdx.y * a.k_per_split;
    const uint32_t c_end = min(a.k, c_begin + a.k_per_split);
    // this wave's contiguous share of the split's columns, in multiples of 4 columns
    const uint32_t ncol4 = (c_end - c_begin + 3u) / 4u;
    const uint32_t per_wave = ((ncol4 + kWaves - 1) / kWaves) * 4u;
    const uint32_t w_begin = min(c_end, c_begin + wave * per_wave);
    const uint32_t w_end = min(c_end, w_begin + per_wave);

    const T *mp = a.m + z * a.m_batch + (row_ok ? row : 0u);
    const T *vp = a.v + z * a.v_batch;

    float4 acc[NRHS];
#pragma unroll
    for (int y = 0; y < NRHS; ++y) acc[y] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (uint32_t cb = w_begin; cb < w_end; cb += 64u) {
        // one coalesced fetch of up to 64 v entries per RHS, broadcast below with v_readlane
        float vv[NRHS];
#pragma unroll
        for (int y = 0; y < NRHS; ++y)
            vv[y] = (cb + lane < w_end && y0 + y < a.nrhs) ? (float)vp[(uint64_t)(y0 + y) * a.ldv + cb + lane] : 0.f;
        const T *col = mp + (uint64_t)cb * a.ldm;
        const uint64_t ld4 = a.ldm; // elements between columns
        if (cb + 64u <= w_end) {
#pragma unroll
            for (int u8 = 0; u8 < 64; u8 += GEMV_NU) {
                float4 mv[GEMV_NU];
#pragma unroll
                for (int u = 0; u < GEMV_NU; ++u) mv[u] = load4s(col + (uint64_t)(u8 + u) * ld4);
#pragma unroll
                for (int u = 0; u < GEMV_NU; ++u)
#pragma unroll
                    for (int y = 0; y < NRHS; ++y) fma4(acc[y], mv[u], readlane_f(vv[y], u8 + u));
            }
        } else {
            // < 64 columns left in this wave's range: the same groups of GEMV_NU loads in flight -- a column past the end re-loads the last
            // valid one and is zeroed (one load at a time here cost the mid-size shapes,
            // whose waves own fewer than 64 columns, up to 30 %: profiles/r03_evidence.md section 10)
            const int rem = (int)(w_end - cb); // wave-uniform, >= 1
            for (int u8 = 0; u8 < rem; u8 += GEMV_NU) {
                float4 mv[GEMV_NU];
#pragma unroll
                for (int u = 0; u < GEMV_NU; ++u) {
                    mv[u] = load4s(col + (uint64_t)min(u8 + u, rem - 1) * ld4); // (the load still issues: a valid address)
                    // ... but a padded slot contributes an exact zero, not (last column) x 0: an Inf or NaN in the last column must not
                    // reach rows through slots the reference kernel never touches (gemv.wgsl:28-65 stops at the last column)
                    if (u8 + u >= rem) mv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < GEMV_NU; ++u)
#pragma unroll
                    for (int y = 0; y < NRHS; ++y) fma4(acc[y], mv[u], readlane_f(vv[y], u8 + u));
            }
        }
    }

#pragma unroll
    for (int y = 0; y < NRHS; ++y) part[wave][y][lane] = acc[y];
    __syncthreads();
    if (wave == 0 && row_ok) {
#pragma unroll
        for (int y = 0; y < NRHS; ++y) {
            if (y0 + y >= a.nrhs) break;
            float4 s = part[0][y][lane];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) {
                float4 p = part[w][y][lane];
                s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
            }
            const uint64_t off = z * a.dst_batch + blockIdx.y * a.dst_split + (uint64_t)(y0 + y) * a.ld_dst + row;
            if (a.part) store4(a.part + off, s); else store4(a.out + off, s);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// T: dst[c] = sum_r m[r, c] v[r],  4 columns per wave, rows of this split
// grid = (column groups of 16, splits, nmats * rhs groups)
// ------------------------------------------------------------------------------------------------------
template <int NRHS, typename T>
__global__ __launch_bounds__(kThreads) void gemv_t_kernel(GemvArgsT<T> a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t rhs_groups = (a.nrhs + kMaxRhs - 1) / kMaxRhs;
    const uint32_t z = blockIdx.z / rhs_groups, y0 = (blockIdx.z % rhs_groups) * kMaxRhs;
    const uint32_t col0 = (blockIdx.x * kWaves + wave) * 4u;
    if (col0 >= a.rows_out) return; // rows_out % 4 == 0: a wave's 4 columns are all in or all out

    const uint32_t r_begin = blockIdx.y * a.k_per_split;
    const uint32_t r_end = min(a.k, r_begin + a.k_per_split);

    const T *mp = a.m + z * a.m_batch + (uint64_t)col0 * a.ldm;
    const T *vp = a.v + z * a.v_batch;
    const uint64_t ld4 = a.ldm; // elements between columns

    float acc[4][NRHS];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int y = 0; y < NRHS; ++y) acc[c][y] = 0.f;

    auto step = [&](uint32_t r) { // r: this lane's first row (multiple of 4), r < r_end
        const T *m4 = mp + r;
        float4 mv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) mv[c] = load4s(m4 + c * ld4);
#pragma unroll
        for (int y = 0; y < NRHS; ++y) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y0 + y < a.nrhs) x = load4(vp + (uint64_t)(y0 + y) * a.ldv + r);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c][y] = fmaf(mv[c].x, x.x, acc[c][y]);
                acc[c][y] = fmaf(mv[c].y, x.y, acc[c][y]);
                acc[c][y] = fmaf(mv[c].z, x.z, acc[c][y]);
                acc[c][y] = fmaf(mv[c].w, x.w, acc[c][y]);
            }
        }
    };

    uint32_t r = r_begin + 4u * lane; // (f16: 8-byte loads of 4 rows; a 16-byte / 8-row variant measured slower: 5.1 vs 5.7 TB/s at 65536 x 4096)
    // TS row-steps per trip: 4*TS matrix loads (+ vector loads) in flight per lane; f16 loads are 8 bytes: twice as many keep the same bytes in flight
    constexpr int TS = (sizeof(T) == 2 ? WG_GEMV_TS_F16 : 1) * GEMV_TS;
    for (; (uint64_t)r + 256u * (TS - 1) < r_end; r += 256u * TS) {
#pragma unroll
        for (int t = 0; t < TS; ++t) step(r + 256u * t);
    }
    for (; r < r_end; r += 256u) step(r);

#pragma unroll
    for (int y = 0; y < NRHS; ++y) {
        if (y0 + y >= a.nrhs) break;
        float4 s;
        s.x = wave_sum(acc[0][y]);
        s.y = wave_sum(acc[1][y]);
        s.z = wave_sum(acc[2][y]);
        s.w = wave_sum(acc[3][y]);
        if (lane == 0) {
            const uint64_t off = z * a.dst_batch + blockIdx.y * a.dst_split + (uint64_t)(y0 + y) * a.ld_dst + col0;
            if (a.part) store4(a.part + off, s); else store4(a.out + off, s);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// T, one right-hand side, the batched-Reduce shape (reduce.hip, reduce_rows4: 6.5 TB/s on the same matrix): a HALF-wave per column, lane p
// takes rows 4 p + 128 i of this split -- 512 contiguous bytes (f32) per half-wave per load, U loads in flight per lane, the vector's
// float4 beside each (the 8 half-waves of a workgroup read the same addresses: cache hits). 8 columns per workgroup.
// grid = (column groups of 8, splits, nmats)
// ------------------------------------------------------------------------------------------------------
#ifndef WG_GEMVT_U
#define WG_GEMVT_U 8
#endif
#ifndef WG_GEMVT_ROT
#define WG_GEMVT_ROT 0 // 1: column c starts its sweep (c * 5) blocks of rows in and wraps around (de-phases the columns' streams)
#endif
typedef _Float16 wg_h8 __attribute__((ext_vector_type(8)));
// E consecutive elements as floats (E = 4: load4 / load4s; E = 8: f16 only, one 16-byte load)
template <int E, typename T> struct RowPiece { float f[E]; };
template <int E, typename T>
__device__ __forceinline__ RowPiece<E, T> piece_stream(const T *p) {
    RowPiece<E, T> o;
    if constexpr (E == 4) { const float4 v = load4s(p); o.f[0] = v.x; o.f[1] = v.y; o.f[2] = v.z; o.f[3] = v.w; }
    else {
        const wg_h8 v = GEMV_NT ? __builtin_nontemporal_load(reinterpret_cast<const wg_h8 *>(p)) : *reinterpret_cast<const wg_h8 *>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) o.f[i] = (float)v[i];
    }
    return o;
}
template <int E, typename T>
__device__ __forceinline__ RowPiece<E, T> piece_cached(const T *p) {
    RowPiece<E, T> o;
    if constexpr (E == 4) { const float4 v = load4(p); o.f[0] = v.x; o.f[1] = v.y; o.f[2] = v.z; o.f[3] = v.w; }
    else {
        const wg_h8 v = *reinterpret_cast<const wg_h8 *>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) o.f[i] = (float)v[i];
    }
    return o;
}
template <typename T, int E, int U = WG_GEMVT_U, int NRHS = 1>
__global__ __launch_bounds__(kThreads) void gemv_t_cols_kernel(GemvArgsT<T> a) {
    const uint32_t col = blockIdx.x * (kThreads / 32) + (threadIdx.x >> 5);
    const uint32_t p = threadIdx.x & 31u;
    const uint32_t z = blockIdx.z;
    if (col >= a.rows_out) return; // whole half-waves leave together; the shuffles below never cross halves
    const uint32_t r_begin = blockIdx.y * a.k_per_split;
    const uint32_t r_end = min(a.k, r_begin + a.k_per_split);
    const T *mp = a.m + z * a.m_batch + (uint64_t)col * a.ldm;
    const T *vp = a.v + z * a.v_batch;
    // U loads in flight per lane per trip. Columns of at most 4 chunks take the U = 4 instantiation (the launcher): half the registers, twice the half-waves per CU -- what
    // a short column lacks in loads per half-wave it gets back in half-waves (f16 GemvTr 1024 x 65536: 55 us in the tail loop, 40 us as a partial trip of U = 8, see the bench line)
    constexpr uint32_t kChunk = 32u * E;   // rows a half-wave covers per load
    constexpr uint32_t kBlock = kChunk * U; // ... per trip
    // NRHS right-hand sides (1, or 2: round 5): every piece of the matrix is multiplied with the same rows of each vector (columns of v, ldv apart; results ld_dst apart)
    float acc[NRHS][E];
#pragma unroll
    for (int y = 0; y < NRHS; ++y)
#pragma unroll
        for (int e = 0; e < E; ++e) acc[y][e] = 0.f;
    const uint32_t nblocks = (r_end - r_begin) / kBlock;
    uint32_t rot = 0;
    if constexpr (WG_GEMVT_ROT) rot = nblocks ? (col * 5u) % nblocks : 0u;
    for (uint32_t i = 0; i < nblocks; ++i) {
        uint32_t j = i + rot;
        if (j >= nblocks) j -= nblocks;
        const uint32_t r = r_begin + j * kBlock + E * p;
        RowPiece<E, T> mv[U], xv[NRHS][U];
#pragma unroll
        for (int u = 0; u < U; ++u) mv[u] = piece_stream<E, T>(mp + r + kChunk * u);
#pragma unroll
        for (int y = 0; y < NRHS; ++y)
#pragma unroll
            for (int u = 0; u < U; ++u) xv[y][u] = piece_cached<E, T>(vp + (uint64_t)y * a.ldv + r + kChunk * u);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int y = 0; y < NRHS; ++y)
#pragma unroll
                for (int e = 0; e < E; ++e) acc[y][e] = fmaf(mv[u].f[e], xv[y][u].f[e], acc[y][e]);
    }
    // what is left of the range (< one trip): its whole chunks as ONE partial trip, all of its loads in flight together (round 5: columns shorter than a trip -- f16 with up to
    // 2047 rows, 8 rows per lane -- ran entirely in the one-load-at-a-time loop below: f16 GemvTr 1024 x 65536 55 us = 2.4 TB/s, the f32 one of the same shape 7.2 TB/s)
    const uint32_t rem_chunks = ((r_end - r_begin) - nblocks * kBlock) / kChunk; // < U, wave-uniform
    if (rem_chunks) {
        const uint32_t r = r_begin + nblocks * kBlock + E * p;
        RowPiece<E, T> mv[U], xv[NRHS][U];
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if ((uint32_t)u < rem_chunks) mv[u] = piece_stream<E, T>(mp + r + kChunk * u);
#pragma unroll
        for (int y = 0; y < NRHS; ++y)
#pragma unroll
            for (int u = 0; u < U - 1; ++u)
                if ((uint32_t)u < rem_chunks) xv[y][u] = piece_cached<E, T>(vp + (uint64_t)y * a.ldv + r + kChunk * u);
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if ((uint32_t)u < rem_chunks) {
#pragma unroll
                for (int y = 0; y < NRHS; ++y)
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[y][e] = fmaf(mv[u].f[e], xv[y][u].f[e], acc[y][e]);
            }
    }
    // ... and the rows behind them (< one chunk): 4 rows per lane, 128 per half-wave (k % 4 == 0: a lane's 4 rows are all in or all out)
    for (uint32_t r = r_begin + nblocks * kBlock + rem_chunks * kChunk + 4u * p; r < r_end; r += 128u) {
        const float4 mv = load4s(mp + r);
#pragma unroll
        for (int y = 0; y < NRHS; ++y) {
            const float4 xv = load4(vp + (uint64_t)y * a.ldv + r);
            acc[y][0] = fmaf(mv.x, xv.x, acc[y][0]);
            acc[y][1] = fmaf(mv.y, xv.y, acc[y][1]);
            acc[y][2] = fmaf(mv.z, xv.z, acc[y][2]);
            acc[y][3] = fmaf(mv.w, xv.w, acc[y][3]);
        }
    }
#pragma unroll
    for (int y = 0; y < NRHS; ++y) {
        float s;
        if constexpr (E == 4) s = (acc[y][0] + acc[y][1]) + (acc[y][2] + acc[y][3]);
        else s = ((acc[y][0] + acc[y][1]) + (acc[y][2] + acc[y][3])) + ((acc[y][4] + acc[y][5]) + (acc[y][6] + acc[y][7]));
#pragma unroll
        for (int sh = 16; sh >= 1; sh >>= 1) s += __shfl_xor(s, sh, 64);
        if (p == 0) {
            const uint64_t off = z * a.dst_batch + blockIdx.y * a.dst_split + (uint64_t)y * a.ld_dst + col;
            if (a.part) a.part[off] = s; else a.out[off] = (T)s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// T with 2 .. 8 right-hand sides: the small-batch product x^T W of a weight matrix stored k x outputs. The vectors are staged into the LDS (as f32) in
// chunks of KC contracted rows -- the whole of k when it fits, which is the common case: then once per workgroup --, and every half-wave streams its
// columns of the matrix exactly like gemv_t_cols_kernel (whole 128-byte lines, U loads in flight) and multiplies each 16-byte piece with the NRHS pieces
// of the vectors read from the LDS: no split of k across workgroups, no combine pass, the matrix read once, and the work is FMAs on the vector unit (the
// few-column Gemm kernels run the same products on the matrix cores at 1/4 .. 1/8 utilisation and pay for it in clock).
// Shape of a workgroup (round 5; before: always 1024 threads x 4 columns per half-wave = 128 columns per trip, which left 2/3 of the chip idle at 11008 outputs):
// THREADS / 32 half-waves x COLS adjacent columns per trip; the launcher picks the pair that gives every CU a workgroup and, when it can, all of them ONE trip
// (gemv_t_lds_plan). COLS > 1 shares each LDS piece of the vectors between COLS matrix pieces (8 right-hand sides with COLS = 1 read 8 x the matrix's bytes
// out of the LDS: 104 of the LDS's 128 bytes per clock at HBM speed).
// grid = (workgroups, 1, nmats), persistent over the column groups g, g + gridDim.x, ...; dynamic LDS = KC * NRHS * 4 bytes.
// ------------------------------------------------------------------------------------------------------
template <int NRHS, typename T, int COLS, int THREADS>
__global__ __launch_bounds__(THREADS) void gemv_t_lds_kernel(GemvArgsT<T> a, uint32_t kc) {
    extern __shared__ __attribute__((aligned(16))) float vs[]; // [y][kc]
    const uint32_t z = blockIdx.z;
    const uint32_t kk = a.k; // % 4 == 0
    const T *vp = a.v + z * a.v_batch;
    const uint32_t p = threadIdx.x & 31u, hw = threadIdx.x >> 5;
    constexpr uint32_t kGroup = (THREADS / 32) * COLS; // columns per workgroup and trip
    const uint32_t nchunks = (kk + kc - 1u) / kc;
    auto stage = [&](uint32_t k0, uint32_t len) { // vectors' rows [k0, k0 + len) -> vs[y][0 .. len), len % 4 == 0
        for (uint32_t idx = threadIdx.x * 4u; idx < len * (uint32_t)NRHS; idx += THREADS * 4u) {
            const uint32_t y = idx / len, r = idx - y * len; // (len % 4 == 0: a thread's 4 entries belong to one vector)
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y < a.nrhs) x = load4(vp + (uint64_t)y * a.ldv + k0 + r);
            *reinterpret_cast<float4 *>(vs + (uint32_t)y * kc + r) = x;
        }
    };
    if (nchunks == 1u) { // staged once, every trip reads it
        stage(0u, kk);
        __syncthreads();
    }
    // rows_out % 4 == 0 and COLS | 4: a half-wave's columns are all in or all out
    for (uint32_t g0 = blockIdx.x * kGroup; g0 < a.rows_out; g0 += gridDim.x * kGroup) {
        const uint32_t c0 = g0 + hw * (uint32_t)COLS;
        const bool live = c0 < a.rows_out;
        const T *mp = a.m + z * a.m_batch + (uint64_t)(live ? c0 : 0u) * a.ldm;
        float acc[COLS][NRHS];
#pragma unroll
        for (int c = 0; c < COLS; ++c)
#pragma unroll
            for (int y = 0; y < NRHS; ++y) acc[c][y] = 0.f;
        for (uint32_t ch = 0; ch < nchunks; ++ch) {
            const uint32_t k0 = ch * kc, len = min(kc, kk - k0);
            if (nchunks > 1u) {
                __syncthreads(); // everyone is done with the previous chunk
                stage(k0, len);
                __syncthreads();
            }
            if (!live) continue; // (after the barriers: a workgroup's half-waves past the last column still take part in them)
            constexpr int U = 4 / COLS; // row-steps in flight per lane: COLS * U = 4 loads of 16 bytes (8 measured: 4096 x 11008 x 4 55 -> 86 us, 4096 x 65536 x 8 178 -> 196)
            uint32_t r = 4u * p;
            for (; r + 128u * (U - 1) < len; r += 128u * U) {
                float4 mv[U][COLS];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int c = 0; c < COLS; ++c) mv[u][c] = load4s(mp + (uint64_t)c * a.ldm + k0 + r + 128u * u);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int y = 0; y < NRHS; ++y) {
                        const float4 x = *reinterpret_cast<const float4 *>(vs + (uint32_t)y * kc + r + 128u * u);
#pragma unroll
                        for (int c = 0; c < COLS; ++c) {
                            acc[c][y] = fmaf(mv[u][c].x, x.x, acc[c][y]); acc[c][y] = fmaf(mv[u][c].y, x.y, acc[c][y]);
                            acc[c][y] = fmaf(mv[u][c].z, x.z, acc[c][y]); acc[c][y] = fmaf(mv[u][c].w, x.w, acc[c][y]);
                        }
                    }
            }
            for (; r < len; r += 128u) { // what is left of the chunk (< U row-steps)
                float4 mv[COLS];
#pragma unroll
                for (int c = 0; c < COLS; ++c) mv[c] = load4s(mp + (uint64_t)c * a.ldm + k0 + r);
#pragma unroll
                for (int y = 0; y < NRHS; ++y) {
                    const float4 x = *reinterpret_cast<const float4 *>(vs + (uint32_t)y * kc + r);
#pragma unroll
                    for (int c = 0; c < COLS; ++c) {
                        acc[c][y] = fmaf(mv[c].x, x.x, acc[c][y]); acc[c][y] = fmaf(mv[c].y, x.y, acc[c][y]);
                        acc[c][y] = fmaf(mv[c].z, x.z, acc[c][y]); acc[c][y] = fmaf(mv[c].w, x.w, acc[c][y]);
                    }
                }
            }
        }
        if (!live) continue;
#pragma unroll
        for (int y = 0; y < NRHS; ++y) {
            float s[COLS];
#pragma unroll
            for (int c = 0; c < COLS; ++c) {
                s[c] = acc[c][y];
#pragma unroll
                for (int sh = 16; sh >= 1; sh >>= 1) s[c] += __shfl_xor(s[c], sh, 64);
            }
            if (p == 0 && (uint32_t)y < a.nrhs) {
                T *o = a.out + z * a.dst_batch + (uint64_t)y * a.ld_dst + c0;
#pragma unroll
                for (int c = 0; c < COLS; ++c) o[c] = (T)s[c];
            }
        }
    }
}

// out[r] = sum_{s < nsplit} partial[s][r] in a FIXED order. partial layout: [z][s][y][rows_out] dense.
// A workgroup covers 4 float4 rows x 64 "split lanes": split lane j adds splits j, j+64, ... ascending (4 independent loads
// in flight per trip), then the 16 split lanes of a wave are folded by a butterfly and the 4 waves ascending through LDS.
// The pass is latency-, not bandwidth-bound: with 16 split lanes and one dependent load per trip it cost 25-50 us when a
// skinny matrix needed ~1000 splits (64 x 2^20: 70 us total against a 42 us stream); this layout keeps it at a few us.
// (Folding the partials in the producing kernel -- last-arriver protocol with device-scope stores and one atomic per
// workgroup -- was measured and rejected: same-address atomics serialise, 64 x 2^20 went to 120 us and 4096^2 from 15 to
// 27 us; it only won 1-2 us on the launch-bound 1024^2 case.)
template <typename T>
__global__ __launch_bounds__(kThreads) void gemv_combine_kernel(const float *__restrict__ partial, uint32_t nsplit, uint32_t rows_out,
                                                                 uint32_t nrhs, T *__restrict__ out, uint32_t ld_out,
                                                                 uint64_t out_batch) {
    __shared__ float4 red[kWaves][4];
    const uint32_t rl = threadIdx.x & 3, sl = threadIdx.x >> 2;
    const uint32_t r4 = blockIdx.x * 4u + rl; // float4 index within a column
    const bool ok = r4 * 4u < rows_out;
    const uint32_t y = blockIdx.y, z = blockIdx.z;
    const uint64_t split_stride = (uint64_t)nrhs * rows_out;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
        const float4 *p = reinterpret_cast<const float4 *>(partial + (uint64_t)z * nsplit * split_stride + (uint64_t)y * rows_out) + r4;
        const uint64_t ss4 = split_stride / 4u;
        uint32_t i = sl;
        for (; i + 192u < nsplit; i += 256u) {
            float4 q0 = p[(uint64_t)i * ss4], q1 = p[(uint64_t)(i + 64u) * ss4], q2 = p[(uint64_t)(i + 128u) * ss4],
                   q3 = p[(uint64_t)(i + 192u) * ss4];
            s.x += q0.x; s.y += q0.y; s.z += q0.z; s.w += q0.w;
            s.x += q1.x; s.y += q1.y; s.z += q1.z; s.w += q1.w;
            s.x += q2.x; s.y += q2.y; s.z += q2.z; s.w += q2.w;
            s.x += q3.x; s.y += q3.y; s.z += q3.z; s.w += q3.w;
        }
        for (; i < nsplit; i += 64u) {
            float4 q = p[(uint64_t)i * ss4];
            s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
    }
    // the 16 split lanes of this wave (lane bits 2..5), fixed butterfly order
#pragma unroll
    for (int st = 4; st <= 32; st <<= 1) {
        s.x += __shfl_xor(s.x, st, 64); s.y += __shfl_xor(s.y, st, 64);
        s.z += __shfl_xor(s.z, st, 64); s.w += __shfl_xor(s.w, st, 64);
    }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) < 4u) red[wave][rl] = s;
    __syncthreads();
    if (threadIdx.x < 4u && ok) {
#pragma unroll
        for (int w = 1; w < kWaves; ++w) {
            float4 q = red[w][rl];
            s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
        store4(out + z * out_batch + (uint64_t)y * ld_out + 4u * r4, s);
    }
}

// ------------------------------------------------------------------------------------------------------
// N, small matrices (a few MiB: launch-bound): ONE kernel, no partials. A workgroup owns 4 RL rows (RL lanes x float4) and all columns:
// its 256 / RL lane groups take columns g, g + 256 / RL, ... with up to 8 loads in flight per lane, are summed across the wave by
// butterfly shuffles and across the four waves through LDS in a fixed order. RL is chosen so that even a 1024-row matrix gives the
// chip >= 128 workgroups (1024 x 1024, BASELINE config 1: RL = 2, every lane's 8 columns in flight at once: one memory round trip
// instead of eight -- the kernel was latency-bound at 32 workgroups of 32 rows).
// gemv_small_rows is shared with the fused Gemv + Reduce below: same summation, same bits.
// ------------------------------------------------------------------------------------------------------
template <int RL, typename T>
__device__ __forceinline__ float4 gemv_small_rows(const T *mp, uint32_t ldm, const T *vp, uint32_t k, float4 (*red)[8]) {
    constexpr uint32_t G = 256u / RL; // column groups of the workgroup
    const uint32_t rl = threadIdx.x & (RL - 1u), g = threadIdx.x / RL;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t c = g;
    for (; c + 7u * G < k; c += 8u * G) { // 8 columns in flight per lane
        float4 mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = load4s(mp + (uint64_t)(c + u * G) * ldm);
#pragma unroll
        for (int u = 0; u < 8; ++u) fma4(acc, mv[u], (float)vp[c + u * G]);
    }
    for (; c < k; c += G) fma4(acc, load4s(mp + (uint64_t)c * ldm), (float)vp[c]);
#pragma unroll
    for (int st = RL; st < 64; st <<= 1) { // the wave's 64 / RL groups, butterfly
        acc.x += __shfl_xor(acc.x, st, 64); acc.y += __shfl_xor(acc.y, st, 64);
        acc.z += __shfl_xor(acc.z, st, 64); acc.w += __shfl_xor(acc.w, st, 64);
    }
    if ((threadIdx.x & 63u) < (uint32_t)RL) red[threadIdx.x >> 6][rl] = acc;
    __syncthreads();
    float4 s = red[0][rl]; // (meaningful for threadIdx.x < RL)
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float4 p = red[w][rl];
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    return s;
}
// rows per workgroup = 4 RL: as few as gives the chip >= 128 workgroups
static inline int gemv_small_rl(uint32_t rows_out) {
    static const int forced = [] { const char *e = getenv("WG_GEMV_SMALL_RL"); return e ? atoi(e) : 0; }(); // experiments: 2 / 4 / 8 lanes x float4 of rows per workgroup
    if (forced == 2 || forced == 4 || forced == 8) return forced;
    return rows_out >= 4096u ? 8 : (rows_out >= 2048u ? 4 : 2);
}

template <int RL, typename T>
__global__ __launch_bounds__(kThreads) void gemv_n_small_kernel(GemvArgsT<T> a) {
    __shared__ float4 red[4][8];
    const uint32_t rl = threadIdx.x & (RL - 1u);
    const uint32_t z = blockIdx.z, y = blockIdx.y;
    const uint32_t row = blockIdx.x * (4u * RL) + 4u * rl;
    const bool row_ok = row < a.rows_out;
    const T *mp = a.m + z * a.m_batch + (row_ok ? row : 0u);
    const T *vp = a.v + z * a.v_batch + (uint64_t)y * a.ldv;
    const float4 s = gemv_small_rows<RL, T>(mp, a.ldm, vp, a.k, red);
    if (threadIdx.x < (uint32_t)RL && row_ok) store4(a.out + z * a.dst_batch + (uint64_t)y * a.ld_dst + row, s);
}

// ------------------------------------------------------------------------------------------------------
// Fused Gemv + Reduce for the launch-bound sizes (reduce.rs:100-113 after gemv.rs:64-137 in ONE launch): the body of
// gemv_n_small_kernel writes y = m v into a scratch vector; the LAST workgroup to finish (one agent-scope release fence + one
// atomic per workgroup -- a few dozen workgroups here; the same protocol lost on the big GEMVs, profiles/r01_evidence.md section 7) then
// folds y in the reference's order: 128 virtual lanes, lane t takes y[t], y[t+128], ... ascending, then the 64..1 tree
// (reduce.wgsl:68-87), mapped onto 32 lanes x float4 exactly like reduce_rows4 -- so the result has the bits of Gemv followed by Reduce.
// ------------------------------------------------------------------------------------------------------
template <int OP, int RL>
__global__ __launch_bounds__(kThreads) void gemv_n_small_reduce_kernel(GemvArgs a, unsigned *__restrict__ counter, float *__restrict__ result) {
    __shared__ float4 red[4][8];
    __shared__ unsigned is_last;
    const uint32_t rl = threadIdx.x & (RL - 1u);
    const uint32_t row = blockIdx.x * (4u * RL) + 4u * rl;
    const bool row_ok = row < a.rows_out;
    const float4 s = gemv_small_rows<RL, float>(a.m + (row_ok ? row : 0u), a.ldm, a.v, a.k, red); // the same summation as gemv_n_small_kernel: y has the bits wg_gemv would have produced
    if (threadIdx.x < (uint32_t)RL && row_ok) {
        *reinterpret_cast<float4 *>(a.out + row) = s;
        __threadfence(); // release: this workgroup's rows of y are visible device-wide (other XCDs' L2 included) before it is counted
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = atomicAdd(counter, 1u);
        is_last = prev == gridDim.x - 1u;
        if (is_last) *counter = 0u; // ready for the next launch on this stream (launches of one context are ordered)
    }
    __syncthreads();
    if (!is_last || threadIdx.x >= 32u) return;
    __threadfence(); // acquire: every other workgroup's y
    const uint32_t p = threadIdx.x, n = a.rows_out; // n % 4 == 0 (vec4 precondition), y is 16-byte aligned scratch
    const float *y = a.out;
    float r[4] = { r_init<OP>(), r_init<OP>(), r_init<OP>(), r_init<OP>() };
    const uint32_t full_rows = n / 128u;
    for (uint32_t q = 0; q < full_rows; ++q) {
        const float4 v = wg_ld_nt(reinterpret_cast<const float4 *>(y) + (uint64_t)q * 32u + p);
        r[0] = r_ws<OP>(r[0], v.x); r[1] = r_ws<OP>(r[1], v.y); r[2] = r_ws<OP>(r[2], v.z); r[3] = r_ws<OP>(r[3], v.w);
    }
    {
        const uint32_t i0 = full_rows * 128u + 4u * p;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
            if (i0 + cc < n) r[cc] = r_ws<OP>(r[cc], __builtin_nontemporal_load(y + i0 + cc));
    }
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) r[cc] = r_red<OP>(r[cc], __shfl_down(r[cc], s, 32));
    }
    r[0] = r_red<OP>(r[0], r[2]);
    r[1] = r_red<OP>(r[1], r[3]);
    r[0] = r_red<OP>(r[0], r[1]);
    if (p == 0) result[0] = r[0];
}

#ifndef GEMV_SMALL
#define GEMV_SMALL 1
#endif
inline uint32_t ceil_div(uint32_t a, uint32_t b) { return a / b + (a % b != 0); }

// blocks along the output, and how finely the contraction must be split to give every CU ~4 workgroups
#ifndef WG_GEMVT_COLS
#define WG_GEMVT_COLS 1 // 0: T with one right-hand side on the 4-columns-per-wave kernel as well (the form before round 3)
#endif
// T with ONE right-hand side runs on gemv_t_cols_kernel (a half-wave per column): measured over 17 shapes x 2 element types against the
// 4-columns-per-wave kernel (tools/gemvtr_sweep.py, profiles/r03_evidence.md section 10): f32 -1...-40 % time (4096^2 16.5 -> 10 us, 65536 x 4096
// 169 -> 162), f16 -5...-45 % (65536 x 4096 95 -> 83 us = 6.5 TB/s) -- except f32 columns exactly 64 KiB apart with k <= 16384 (16384 x 2048 ...
// x 16384: +5...+13 %; 15360 and 17408 rows are fine, f16 and longer columns at that stride are fine), which stay on the old kernel.
// gemv_t_cols_kernel with 4 instead of 8 loads in flight per lane (half the registers, twice the half-waves per CU): columns of at most 8 chunks (a chunk: 32 lanes x 16 bytes),
// and up to 31 chunks when the 8-deep form would end in a partial trip. Measured (tools/gemvtr_sweep.py, f16, us): 1024 x 65536 55 -> 20 (vendor 22), 1280 x 32768 25.4 -> 13.8,
// 3072 x 65536 74.2 -> 57.0; whole trips of 8 stay (4096 x 65536 f16 75.0 against 79.0, 2048 x 65536 f32 76.5 against 79.5).
static bool t_cols_u4(uint32_t k_per_split, uint32_t e) {
    const uint32_t chunks = k_per_split / (32u * e);
    return chunks <= 8u || (chunks < 32u && chunks % 8u != 0);
}
// GemvTr with TWO right-hand sides on the half-wave-per-column kernel's 2-vector form (gemv_t_cols_kernel<.., 4, 2>, round 5): every piece of the matrix meets the same rows
// of both vectors, no LDS, no barrier. Measured against what took these shapes before (tools/misc_sweep.py with MISC_GEMV_SHAPES, us, before -> after | vendor) -- f16 (the
// 4-columns-per-wave kernel): 4096^2 14.0 -> 7.7 | 18, 4096 x 11008 24.0 -> 18.5 | 18.5, 11008 x 4096 23.4 -> 18.1 | 19, 8192^2 36.3 -> 25.0 | 22-27, 65536 x 4096 111 -> 88 | 108,
// 16384^2 104 -> 91 | 98, but few outputs lose (32768 x 1536 28.8 -> 38.7): from 2048 outputs on. f32 (the vectors-in-LDS kernel): only where that kernel's one barrier per chunk
// shows -- 4096 x 11008 41.8 -> 33.3 | 30, 4096^2 12.8 -> 12.0 -- and worse elsewhere (16384^2 163 -> 188, 11008 x 4096 32.3 -> 34.7): contractions of 2049 .. 8192 rows onto
// 4096 .. 16384 outputs.
#ifndef WG_GEMVT_COLS2
#define WG_GEMVT_COLS2 1
#endif
template <typename T>
static bool uses_t_cols2(bool trans, uint32_t nrhs, uint32_t rows_out, uint32_t k) {
    if (!WG_GEMVT_COLS2 || !trans || nrhs != 2u) return false;
    if (sizeof(T) == 2) return rows_out >= 2048u;
    return k > 2048u && k <= 8192u && rows_out >= 4096u && rows_out <= 16384u;
}
template <typename T>
static bool uses_t_cols(bool trans, uint32_t nrhs, uint32_t k, uint32_t ldm) {
    if (!WG_GEMVT_COLS || !trans || nrhs != 1) return false;
    if (sizeof(T) == 4 && ldm % 16384u == 0 && k <= 16384u) return false;
    return true;
}
template <typename T>
static uint32_t plan_nsplit(int cus, bool trans, uint32_t rows_out, uint32_t k, uint32_t nrhs, uint32_t gz, uint32_t ldm) {
    const bool cols = uses_t_cols<T>(trans, nrhs, k, ldm) || uses_t_cols2<T>(trans, nrhs, rows_out, k);
    const uint32_t gx = cols ? ceil_div(rows_out, 8u) : trans ? ceil_div(rows_out, 4u * kWaves) : ceil_div(rows_out, 256u);
    const uint32_t min_k_per_split = trans ? 2048u : 64u; // T: >= 8 row-steps per lane; N: >= 16 columns per wave
    const uint64_t blocks_xy = (uint64_t)gx * gz;
    const uint32_t max_split = k == 0 ? 1u : ceil_div(k, min_k_per_split);
    // ~4 workgroups per CU (measured at 1 / 2 / 4 per CU: GemvTr 65536 x 4096 5.4 / 6.0 / 6.4 TB/s; Gemv 4096 x 11008 43 / - / 33 us),
    // except a tall matrix with a short contraction whose row blocks alone give every CU a workgroup: splitting 65536 x 256 four ways
    // costs 18 us with the combine pass against 10 us unsplit.
    // N: matrices of 256 MiB and more stream best with 2 workgroups per CU and four times the columns per workgroup (4096 x 32768 f32: 103 -> 83 us,
    // 16384^2: 167 -> 158, 4096 x 65536: 165.9 -> 162.5; f16 alike), smaller ones with 4 (tools/gemvtr_sweep.py with SWEEP_N=1, r03 evidence section 10)
    // (round 5: tall matrices from 128 MiB on as well -- 32768 x 1536 f32 39.3 -> 35.9 us, 16384 x 2048 30.4 -> 27.4; not the squarer ones: 8192 x 4096 23.1 -> 27.4, 11008 x 4096 31.2 -> 33.4)
    const uint64_t n_bytes = (uint64_t)rows_out * k * sizeof(T);
    const bool big_n = !trans && (n_bytes >= (256ull << 20) || (sizeof(T) == 4 && n_bytes >= (128ull << 20) && rows_out >= 16384u));
    const uint32_t per_cu = cols ? 2u : trans ? 4u : big_n ? 2u : 4u; // (the half-wave-per-column kernel: 2 / 4 / 8 per CU measured the same within 2 %; fewer splits, smaller combine)
    uint32_t want = blocks_xy >= (uint64_t)cus * per_cu ? 1u : ceil_div((uint32_t)cus * per_cu, (uint32_t)blocks_xy);
    if (!trans && blocks_xy >= (uint64_t)cus && k <= 1024u) want = 1u;
    if (trans && !cols && blocks_xy >= 2ull * (uint64_t)cus) want = 1u; // T with >= 2 workgroups per CU already: 8192 x 8192 47 us unsplit, 57 split in two + combine
    uint32_t nsplit = want > max_split ? max_split : want;
    if (nsplit < 1u) nsplit = 1u;
    if (nsplit > 65535u) nsplit = 65535u;
    if (nrhs > 65535u) nsplit = 1u; // the combine pass puts the right-hand sides on grid.y; that many columns fill the chip unsplit
    return nsplit;
}
// wgk_gemv runs launch-bound Gemvs as ONE kernel without partials (gemv_n_small_kernel); the fused Gemv+Reduce must pick exactly those
static bool uses_small_kernel(int cus, bool trans, uint32_t rows_out, uint32_t k, uint32_t nrhs, uint32_t nsplit) {
    return GEMV_SMALL && !trans && nsplit > 1 && (uint64_t)rows_out * k <= (4ull << 20) && rows_out >= 128u && nrhs <= 65535u;
}

} // namespace

template <typename T>
static int gemv_launch(wg_ctx *ctx, bool trans, uint32_t rows_out, uint32_t k, uint32_t nrhs, uint32_t nmats, T *out, uint32_t out_ld, uint64_t out_batch,
                       wgk_mat m, wgk_mat v) {
    const int cus = ctx->compute_units > 0 ? ctx->compute_units : 256;
    const uint32_t rhs_groups = ceil_div(nrhs, kMaxRhs);
    const uint64_t gz64 = (uint64_t)nmats * rhs_groups;
    if (gz64 > 65535) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemv: nmats * ceil(nrhs/8) = %llu exceeds 65535", (unsigned long long)gz64);
    const uint32_t gz = (uint32_t)gz64;

    const bool t_cols = uses_t_cols<T>(trans, nrhs, k, m.ld) || uses_t_cols2<T>(trans, nrhs, rows_out, k);
    const uint32_t gx = t_cols ? ceil_div(rows_out, 8u) : trans ? ceil_div(rows_out, 4u * kWaves) : ceil_div(rows_out, 256u);
    uint32_t nsplit = plan_nsplit<T>(cus, trans, rows_out, k, nrhs, gz, m.ld);
    uint32_t k_per_split = k == 0 ? 4u : ceil_div(ceil_div(k, nsplit), 4u) * 4u; // vec4 granularity
    nsplit = k == 0 ? 1u : ceil_div(k, k_per_split);

    GemvArgsT<T> a;
    a.m = (const T *)m.ptr; a.ldm = m.ld; a.m_batch = m.batch;
    a.v = (const T *)v.ptr; a.ldv = v.ld; a.v_batch = v.batch;
    a.rows_out = rows_out; a.k = k; a.nrhs = nrhs; a.k_per_split = k_per_split;
    a.out = out;
    a.part = nullptr;
    if (nsplit > 1) { // f32 partial sums whatever the element type
        size_t bytes = (size_t)nmats * nsplit * nrhs * rows_out * sizeof(float);
        void *ws = nullptr;
        if (int rc = wg_ctx_workspace(ctx, bytes, &ws)) return rc;
        a.part = (float *)ws;
        a.ld_dst = rows_out;
        a.dst_split = (uint64_t)nrhs * rows_out;
        a.dst_batch = (uint64_t)nsplit * nrhs * rows_out;
    } else {
        a.ld_dst = out_ld;
        a.dst_split = 0;
        a.dst_batch = out_batch;
    }

    // launch-bound sizes: one kernel without partials beats split + combine (1024 x 1024: 12.6 -> ~9 us per eager dispatch)
    if (uses_small_kernel(cus, trans, rows_out, k, nrhs, nsplit)) {
        a.part = nullptr; a.ld_dst = out_ld; a.dst_split = 0; a.dst_batch = out_batch;
        const int rl = gemv_small_rl(rows_out);
        const dim3 sg(ceil_div(rows_out, 4u * (uint32_t)rl), nrhs, nmats);
        if (rl == 8) hipLaunchKernelGGL((gemv_n_small_kernel<8, T>), sg, dim3(kThreads), 0, ctx->stream, a);
        else if (rl == 4) hipLaunchKernelGGL((gemv_n_small_kernel<4, T>), sg, dim3(kThreads), 0, ctx->stream, a);
        else hipLaunchKernelGGL((gemv_n_small_kernel<2, T>), sg, dim3(kThreads), 0, ctx->stream, a);
        WG_HIP_TRY(hipGetLastError());
        return WG_OK;
    }
    const dim3 grid(gx, nsplit, gz), block(kThreads);
    // kMaxRhs RHS columns per group (grid.z); the kernel template is the per-pass register tile: the smallest of 1, 2, 4, 8 that
    // holds min(nrhs, 8) columns (a group of 3 uses tile 4, of 5..7 tile 8)
    const uint32_t per_group = nrhs < (uint32_t)kMaxRhs ? nrhs : (uint32_t)kMaxRhs;
    const int tile = per_group > 4 ? 8 : (per_group > 2 ? 4 : (int)per_group);
    if (t_cols) {
        // f16: 16-byte loads (8 rows per lane) where every column, split and batch keeps them aligned; 8-byte loads (the view contract) otherwise
        bool wide = false;
        if constexpr (sizeof(T) == 2)
            wide = (uintptr_t)a.m % 16 == 0 && (uintptr_t)a.v % 16 == 0 && a.ldm % 8 == 0 && a.k_per_split % 8 == 0 && (nmats == 1 || (a.m_batch % 8 == 0 && a.v_batch % 8 == 0)) &&
                   (nrhs == 1 || a.ldv % 8 == 0);
        if (nrhs == 2) { // (uses_t_cols2)
            if constexpr (sizeof(T) == 2) {
                if (wide) hipLaunchKernelGGL((gemv_t_cols_kernel<T, 8, 4, 2>), grid, block, 0, ctx->stream, a);
                else hipLaunchKernelGGL((gemv_t_cols_kernel<T, 4, 4, 2>), grid, block, 0, ctx->stream, a);
            } else hipLaunchKernelGGL((gemv_t_cols_kernel<T, 4, 4, 2>), grid, block, 0, ctx->stream, a);
        } else if constexpr (sizeof(T) == 2) {
            if (wide && t_cols_u4(a.k_per_split, 8u)) hipLaunchKernelGGL((gemv_t_cols_kernel<T, 8, 4>), grid, block, 0, ctx->stream, a);
            else if (wide) hipLaunchKernelGGL((gemv_t_cols_kernel<T, 8>), grid, block, 0, ctx->stream, a);
            else if (t_cols_u4(a.k_per_split, 4u)) hipLaunchKernelGGL((gemv_t_cols_kernel<T, 4, 4>), grid, block, 0, ctx->stream, a);
            else hipLaunchKernelGGL((gemv_t_cols_kernel<T, 4>), grid, block, 0, ctx->stream, a);
        } else if (t_cols_u4(a.k_per_split, 4u)) hipLaunchKernelGGL((gemv_t_cols_kernel<T, 4, 4>), grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL((gemv_t_cols_kernel<T, 4>), grid, block, 0, ctx->stream, a);
    }
    else if (trans) {
        if (tile == 1) hipLaunchKernelGGL((gemv_t_kernel<1, T>), grid, block, 0, ctx->stream, a);
        else if (tile == 2) hipLaunchKernelGGL((gemv_t_kernel<2, T>), grid, block, 0, ctx->stream, a);
        else if (tile == 4) hipLaunchKernelGGL((gemv_t_kernel<4, T>), grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL((gemv_t_kernel<8, T>), grid, block, 0, ctx->stream, a);
    } else {
        if (tile == 1) hipLaunchKernelGGL((gemv_n_kernel<1, T>), grid, block, 0, ctx->stream, a);
        else if (tile == 2) hipLaunchKernelGGL((gemv_n_kernel<2, T>), grid, block, 0, ctx->stream, a);
        else if (tile == 4) hipLaunchKernelGGL((gemv_n_kernel<4, T>), grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL((gemv_n_kernel<8, T>), grid, block, 0, ctx->stream, a);
    }
    WG_HIP_TRY(hipGetLastError());
    if (nsplit > 1) {
        hipLaunchKernelGGL(gemv_combine_kernel<T>, dim3(ceil_div(rows_out / 4u, 4u), nrhs, nmats), block, 0, ctx->stream, a.part, nsplit, rows_out, nrhs, out,
                           out_ld, out_batch);
        WG_HIP_TRY(hipGetLastError());
    }
    return WG_OK;
}

// 3 .. 8 right-hand sides on a matrix that is not launch-bound: the GEMV kernels re-read the vectors once per matrix piece (GemvTr) or give every
// right-hand side its own accumulators (f16 Gemv), and fall well behind one pass on the matrix cores -- measured against both and against the
// vendor's skinny GEMMs (tools/misc_sweep.py, profiles/r03_evidence.md section 12): f16 GemvTr 65536 x 4096 x 8 373 -> 135 us, f32 GemvTr 4096^2 x 8
// 33 -> 18 us, f16 Gemv 65536 x 4096 x 8 138 -> 123 us. Kept on the GEMV kernels: 2 right-hand sides, matrices under 48 MiB (one launch instead
// of several), f32 Gemv (the N kernel streams 8 right-hand sides at 6.2 TB/s), and f32 GemvTr with >= 8 outputs per contracted row (212 vs 279 us).
static bool few_rhs_as_gemm(bool trans, bool f16, uint32_t rows_out, uint32_t k, uint32_t nrhs, uint32_t es) {
    const uint64_t bytes = (uint64_t)rows_out * k * es;
    if (nrhs < 3u || bytes < ((nrhs >= 7u ? 16ull : 48ull) << 20)) return false; // (7-8 right-hand sides pay from 16 MiB: 4096^2 f16 x 8 26 -> 19 us)
    if (trans) return f16 || (uint64_t)rows_out < 8ull * k;
    // f32 Gemv: since the few-column Gemm kernel multiplies N <= 16 on 16-wide MFMAs (round 5) it is ahead of the 8-accumulator N kernel on everything but a few
    // rows with a long contraction (11008 x 4096 x 4: 36.4 -> 33.2 us, 65536 x 4096 x 8: 187 -> 176, 8192^2 x 3: 47.7 -> 42.9; 4096 x 65536 x 8: 173 stays, Gemm 180)
    // (With the non-temporal hint on its streamed pieces -- gemm_f32_skinny.hip tr_dma_streamed, round 5 -- the Gemm form of 4096 x 65536 x 8 takes 160 us in a replayed
    // command buffer and 175 +- 14 us, at 1.79 GHz, as eager dispatches in bench.py, against the N kernel's steady 173 at 2.39 GHz: it stays on the N kernel.)
    return f16 || (uint64_t)k < 8ull * rows_out;
}

#ifndef WG_GEMVT_LDS
#define WG_GEMVT_LDS 1
#endif
// GemvTr with 2 .. 8 right-hand sides on gemv_t_lds_kernel: shape of the workgroups, chunk of k in the LDS, grid.
struct TLdsPlan {
    int cols, threads;   // columns per half-wave, threads per workgroup
    uint32_t kc, grid;   // contracted rows per LDS chunk (== k: one chunk), workgroups
    size_t lds;
};
static TLdsPlan gemv_t_lds_plan(uint32_t cus, uint32_t rows_out, uint32_t k, uint32_t tile) {
    TLdsPlan pl;
    // the vectors whole when k x tile floats fit 128 KiB (then they are staged once per workgroup), else chunks of 64 KiB (two workgroups per CU)
    const uint32_t whole = (128u << 10) / (tile * 4u), chunk = (64u << 10) / (tile * 4u);
    pl.kc = k <= whole ? k : chunk;
    pl.lds = (size_t)pl.kc * tile * sizeof(float);
    static const int shapes[5][2] = { { 4, 1024 }, { 2, 1024 }, { 1, 1024 }, { 1, 512 }, { 1, 256 } }; // 128, 64, 32, 16, 8 columns per trip
    double best = 1e30;
    pl.cols = 1; pl.threads = 256; pl.grid = 1;
    for (const auto &sh : shapes) {
        const uint32_t per_wg = (uint32_t)sh[0] * (uint32_t)sh[1] / 32u;
        const uint32_t groups = ceil_div(rows_out, per_wg);
        uint32_t per_cu = (uint32_t)((160u << 10) / (pl.lds ? pl.lds : 1));
        if (per_cu > 2048u / (uint32_t)sh[1]) per_cu = 2048u / (uint32_t)sh[1];
        if (per_cu < 1u) per_cu = 1u;
        const uint32_t slots = cus * per_cu, grid = groups < slots ? groups : slots;
        const uint32_t trips = ceil_div(groups, grid);
        // relative time: work per workgroup slot in column-trips, against an even share; a grid that leaves CUs without a workgroup pays for them;
        // narrower shapes pay a little for re-reading the vectors from the LDS per column (COLS = 1) and for staging them per workgroup
        double f = (double)trips * grid / (double)groups;
        if (groups < cus) f *= (double)cus / groups;
        // (staging: tile floats of the vectors per contracted row against per_wg floats of the matrix, from the L2; once per workgroup when k is one chunk)
        f *= 1.0 + 0.02 * (4 - sh[0]) + 0.5 * (double)tile / per_wg * (k > pl.kc ? 1.0 : 1.0 / trips);
        if (f < best) { best = f; pl.cols = sh[0]; pl.threads = sh[1]; pl.grid = grid; }
    }
    return pl;
}
template <typename T>
static int gemv_t_lds_launch(wg_ctx *ctx, uint32_t rows_out, uint32_t k, uint32_t nrhs, uint32_t nmats, T *out, uint32_t out_ld, uint64_t out_batch, wgk_mat m, wgk_mat v) {
    GemvArgsT<T> a;
    a.m = (const T *)m.ptr; a.ldm = m.ld; a.m_batch = m.batch;
    a.v = (const T *)v.ptr; a.ldv = v.ld; a.v_batch = v.batch;
    a.rows_out = rows_out; a.k = k; a.nrhs = nrhs; a.k_per_split = k;
    a.out = out; a.part = nullptr; a.ld_dst = out_ld; a.dst_split = 0; a.dst_batch = out_batch;
    const int tile = nrhs > 4 ? 8 : (nrhs > 2 ? 4 : 2);
    const uint32_t cus = (uint32_t)(ctx->compute_units > 0 ? ctx->compute_units : 256);
    const TLdsPlan pl = gemv_t_lds_plan(cus, rows_out, k, (uint32_t)tile);
    const dim3 grid(pl.grid, 1, nmats);
    // hipFuncAttributeMaxDynamicSharedMemorySize once per context (= per device and stream) and instantiation: bit = 5 * tile index + shape index
#define WG_T_LDS(NR, COLS, THREADS, SHAPE)                                                                                                     \
    do {                                                                                                                                       \
        const uint32_t bit = 1u << (5 * (NR == 8 ? 0 : NR == 4 ? 1 : 2) + SHAPE);                                                              \
        if (!(ctx->lds_attr_bits & bit)) {                                                                                                     \
            WG_HIP_TRY(hipFuncSetAttribute((const void *)gemv_t_lds_kernel<NR, T, COLS, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)); \
            ctx->lds_attr_bits |= bit;                                                                                                         \
        }                                                                                                                                      \
        hipLaunchKernelGGL((gemv_t_lds_kernel<NR, T, COLS, THREADS>), grid, dim3(THREADS), pl.lds, ctx->stream, a, pl.kc);                     \
    } while (0)
#define WG_T_LDS_SHAPES(NR)                                                            \
    do {                                                                               \
        if (pl.cols == 4) WG_T_LDS(NR, 4, 1024, 0);                                    \
        else if (pl.cols == 2) WG_T_LDS(NR, 2, 1024, 1);                               \
        else if (pl.threads == 1024) WG_T_LDS(NR, 1, 1024, 2);                         \
        else if (pl.threads == 512) WG_T_LDS(NR, 1, 512, 3);                           \
        else WG_T_LDS(NR, 1, 256, 4);                                                  \
    } while (0)
    if (tile == 8) WG_T_LDS_SHAPES(8); else if (tile == 4) WG_T_LDS_SHAPES(4); else WG_T_LDS_SHAPES(2);
#undef WG_T_LDS_SHAPES
#undef WG_T_LDS
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}
// Where it runs (f32 only: f16 gains nothing over the f16 Gemm kernels, 120 vs 111 us at 4096 x 65536 x 8), measured with tools/misc_sweep.py
// (profiles/r05_evidence.md section 5):
//   * from 128 outputs per CU on, the vectors whole in the LDS: 4096 x 65536 x 8 178 us against 212 on the
//     4-columns-per-wave kernel and 279 on the few-column Gemm kernel (vendor 178);
//   * TWO right-hand sides from 8 outputs per CU on (the narrow workgroup shapes of round 5): 4096^2 x 2 18.2 -> 12.8 us (vendor 18.9);
//   * NOT 3 .. 8 right-hand sides on fewer outputs: every workgroup shape loses to the matrix-core path there (4096 x 11008 x 4: 55 us on 344 workgroups
//     of 32 columns, 86 with twice the loads in flight, against 40 on the few-column Gemm kernel; 4096^2 x 8 26.5 against 18.1) -- a column is 16 KiB,
//     each half-wave's whole life is a few dependent round trips behind the staging of the vectors.
static bool uses_t_lds(const wg_ctx *ctx, bool trans, uint32_t rows_out, uint32_t k, uint32_t nrhs, uint32_t nmats, uint32_t es) {
    if (!WG_GEMVT_LDS || !trans || es != 4u || nrhs < 2u || nrhs > 8u || nmats > 65535u) return false;
    if (!ctx->tuning[WG_TUNE_GEMVT_LDS] && uses_t_cols2<float>(trans, nrhs, rows_out, k)) return false; // (two right-hand sides on the column kernel's 2-vector form)
    const uint32_t forced = (uint32_t)ctx->tuning[WG_TUNE_GEMVT_LDS], min_cols = forced ? forced : 128u; // (forced: tests, experiments -- wg_ctx_set_tuning)
    const uint32_t cus = (uint32_t)(ctx->compute_units > 0 ? ctx->compute_units : 256);
    const uint32_t tile = nrhs > 4u ? 8u : (nrhs > 2u ? 4u : 2u);
    if (rows_out % 4u || k % 4u || k < 128u) return false;
    const bool whole = (uint64_t)k * tile * 4u <= (128u << 10);                       // one chunk: staged once per workgroup
    const bool chunks_ok = whole || (uint64_t)k * tile * 4u <= 4ull * (64u << 10);     // at most 4 chunks (two barriers per chunk and trip)
    if ((uint64_t)rows_out >= (uint64_t)min_cols * cus) return forced ? chunks_ok : whole;
    return nrhs == 2u && whole && (uint64_t)rows_out >= 8ull * cus;
}

// (Round 5 also carried the half-wave-per-column shape of gemv_t_cols_kernel over to 2 .. 8 right-hand sides -- four adjacent columns per half-wave, the
// vectors re-read through the L1 -- and measured it slower than every path above on all eight sweep cases (4096^2 x 4: 35.8 us against 18.3 on the few-column
// Gemm kernel; 4096 x 11008 x 4: 47 against 39): the re-reads cost the L1 as much as the matrix itself. Removed; profiles/r05_evidence.md section 5.)
int wgk_gemv(wg_ctx *ctx, bool trans, wg_dtype dtype, uint32_t rows_out, uint32_t k, uint32_t nrhs, uint32_t nmats,
             void *out, uint32_t out_ld, uint64_t out_batch, wgk_mat m, wgk_mat v) {
    if (rows_out == 0 || nrhs == 0 || nmats == 0) return WG_OK;
    if (uses_t_lds(ctx, trans, rows_out, k, nrhs, nmats, dtype == WG_F16 ? 2u : 4u)) {
        return gemv_t_lds_launch<float>(ctx, rows_out, k, nrhs, nmats, (float *)out, out_ld, out_batch, m, v);
    }
    if (dtype == WG_F16) {
        // f16 (extension; the reference kernel is f32, gemv.wgsl:9-14): the same HBM-bound kernels on f16 elements (8-byte loads of 4 rows,
        // f32 accumulation in the same order, one rounding at the store; split partials stay f32). More than 8 right-hand sides are a
        // Gemm with few columns: the f16 Gemm kernels take any column count (one pass over the matrix on the matrix cores).
        if (nrhs > (uint32_t)kMaxRhs || few_rhs_as_gemm(trans, true, rows_out, k, nrhs, 2u))
            return wgk_gemm_f16(ctx, trans, rows_out, nrhs, k, nmats, (__half *)out, out_ld, out_batch, m, v, 1.f, 0.f);
        return gemv_launch<_Float16>(ctx, trans, rows_out, k, nrhs, nmats, (_Float16 *)out, out_ld, out_batch, m, v);
    }
    // 9 .. 64 right-hand sides are a Gemm with few columns: one pass over the matrix on the matrix cores (gemm_f32_skinny.hip) instead of
    // one GEMV pass per 8 columns. (The 32-bit DMA offsets of that kernel must suffice for both operands, in both variants.)
    if ((nrhs > (uint32_t)kMaxRhs || few_rhs_as_gemm(trans, false, rows_out, k, nrhs, 4u)) && nrhs <= 64u && rows_out >= 512u && k >= 128u &&
        (uint64_t)m.ld * 32u * 4u < (1ull << 31) && (uint64_t)v.ld * 64u * 4u < (1ull << 31))
        return wgk_gemm_f32_skinny(ctx, trans, rows_out, nrhs, k, nmats, (float *)out, out_ld, out_batch, m, v, 1.f, 0.f);
    return gemv_launch<float>(ctx, trans, rows_out, k, nrhs, nmats, (float *)out, out_ld, out_batch, m, v);
}

// One launch for result = reduce(op, m v) when the Gemv is launch-bound (the single-kernel shape family of wgk_gemv); WG_ERR_UNSUPPORTED
// tells the caller to run Gemv and Reduce as two launches. `y` is a scratch vector of rows_out floats, `counter` a zeroed device word.
int wgk_gemv_small_reduce(wg_ctx *ctx, int op, uint32_t rows_out, uint32_t k, float *y, wgk_mat m, wgk_mat v, unsigned *counter, float *result) {
    const int cus = ctx->compute_units > 0 ? ctx->compute_units : 256;
    if (k < 4u || !uses_small_kernel(cus, false, rows_out, k, 1, plan_nsplit<float>(cus, false, rows_out, k, 1, 1, m.ld))) return WG_ERR_UNSUPPORTED;
    GemvArgs a;
    a.m = (const float *)m.ptr; a.ldm = m.ld; a.m_batch = 0;
    a.v = (const float *)v.ptr; a.ldv = v.ld; a.v_batch = 0;
    a.rows_out = rows_out; a.k = k; a.nrhs = 1; a.k_per_split = k;
    a.out = y; a.part = nullptr; a.ld_dst = rows_out; a.dst_split = 0; a.dst_batch = 0;
    const int rl = gemv_small_rl(rows_out);
    const dim3 grid(ceil_div(rows_out, 4u * (uint32_t)rl)), block(kThreads);
#define WG_SMALL_REDUCE(OP)                                                                                                          \
    do {                                                                                                                             \
        if (rl == 8) hipLaunchKernelGGL((gemv_n_small_reduce_kernel<OP, 8>), grid, block, 0, ctx->stream, a, counter, result);        \
        else if (rl == 4) hipLaunchKernelGGL((gemv_n_small_reduce_kernel<OP, 4>), grid, block, 0, ctx->stream, a, counter, result);   \
        else hipLaunchKernelGGL((gemv_n_small_reduce_kernel<OP, 2>), grid, block, 0, ctx->stream, a, counter, result);               \
    } while (0)
    switch (op) {
    case R_MIN: WG_SMALL_REDUCE(R_MIN); break;
    case R_MAX: WG_SMALL_REDUCE(R_MAX); break;
    case R_SUM: WG_SMALL_REDUCE(R_SUM); break;
    case R_PROD: WG_SMALL_REDUCE(R_PROD); break;
    default: WG_SMALL_REDUCE(R_SQNORM); break;
    }
#undef WG_SMALL_REDUCE
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}
