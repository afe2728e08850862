// f32 Gemm: out = m1 * m2 (NN) or m1^T * m2 (TN), column-major, batched   (wgebra gemm.wgsl:28-200)
//
// Bound: MFMA (f32-in/f32-acc matrix cores, 157.3 TFLOP/s peak = 64 FLOP/clk/SIMD).  v_mfma_f32_32x32x2_f32 is an
// exact k-ordered fmaf chain (guide: cdna_hip_programming.md section 3), so the result is a plain f32 dot product in a
// blocked order; parity with the WGSL orders is tolerance-based.
//
// Tiling (one workgroup = 256 threads = 4 waves, 2(M) x 2(N)):
//   block tile 256(M) x 128(N) x 16(K); wave tile 128 x 64 = 4 x 2 MFMA tiles of 32x32 -> 128 accumulator VGPRs;
//   2 workgroups per CU (launch_bounds(256,2)), LDS 2 x (16 KiB A + 8 KiB B) = 48 KiB per workgroup.
//
// Column-major operands and the MFMA operand shape (lane l supplies ONE scalar: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]):
//   * B (K x N, k contiguous) is staged as Bs[n][16 k] -- the global layout, 16 B per lane straight in. A lane reads a
//     float4 = 4 consecutive k of its column n; the k of MFMA number s is component s. Because any k permutation of a
//     dot product is legal, the two half-waves (h = l>>5) take k-chunks h and 2+h... i.e. chunk (2*ks + h).
//     The 64-byte rows would 4-way conflict on ds_read_b128, so the 16-byte chunk index is XOR-swizzled with (n>>2)&3.
//   * NN: A (M x K, m contiguous) is staged as As[k][256 m] -- again the global layout. A lane reads the float4 of
//     rows 4i..4i+3 at one k: its 4 components feed 4 DIFFERENT M-tiles, so M-tile t holds rows {4i + t}. That row
//     interleave is undone for free in the epilogue: for one accumulator group the 4 tiles x 4 registers of a lane
//     are 16 consecutive rows of C, written as four float4 stores.
//   * TN: m1 is K x M, so op(A)[m][k] is k contiguous like B: As[m][16 k] with the same swizzle, tiles are plain
//     32-row blocks.
// Pipeline (interior tiles): LDS-DMA into a 3-slot ring, two k-tiles ahead, counted vmcnt, barrier in the middle of a run
// of MFMAs, fragments of the next tile prefetched across it (gemm_f32_tile_dma below): 140 TFLOP/s at 4096^3 = 89 % of
// peak, 93.5 % MFMA-cycle utilisation. With 2 workgroups per CU the partner wave on each SIMD covers what is left.
// Edge tiles (gemm_f32_tile<.., true>): global -> registers -> LDS (double buffer), every global access predicated at
// float4 granularity (rows/cols/K are multiples of 4 by the vec4 precondition), out-of-range operands zero-filled, so
// any M, N, K % 4 == 0 is accepted.
// Workgroup ids are remapped so that each XCD (private L2) works on a contiguous band of tiles.
#include "wg_internal.hpp"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
#ifndef WG_ABLATE
#define WG_ABLATE 0 // timing experiments only (results are garbage): 2 = no global loads / LDS stores, 4 = no LDS reads
#endif

constexpr int BM = 256, BN = 128, BK = 16;
constexpr int kThreads = 256;
constexpr int A_TILE = BM * BK; // floats
constexpr int B_TILE = BN * BK;

struct GemmArgs {
    const float *a; uint32_t lda; uint64_t a_batch;
    const float *b; uint32_t ldb; uint64_t b_batch;
    float *c; uint32_t ldc; uint64_t c_batch;
    uint32_t M, N, K;
    uint32_t tiles_m, tiles_n;
    uint32_t dma_ok; // leading dimensions small enough for 32-bit byte offsets within a tile
    uint32_t nsplit, k_per_split; // split-K: grid.y = nmats * nsplit; c then points at the f32 slabs [z][s][N][M]
    float alpha, beta;            // out = alpha * acc + beta * out (wg_gemm_ex); (1, 0) in split mode (the reduce kernel applies them)
    // "tail split" launches (tile quantisation, see wgk_gemm_f32): tile id = tile_base + blockIdx.x; with tail_tiles > 0 workgroup
    // (tile, split) writes its partial tile densely to c[(split * tail_tiles + blockIdx.x) * BM * BN + col_local * BM + row_local]
    uint32_t tile_base, tail_tiles;
    float *out_c; uint32_t out_ldc; float out_alpha, out_beta; // the real output, for gemm_f32_tail_reduce
    // Batches with the tail split (round 6): flat_tiles = tiles per matrix; the launch's ids then run through the matrices' tiles in turn (id = matrix * flat_tiles + tile,
    // grid.y = the K split only), so that the full waves and the cut-up leftover are those of the WHOLE batch. 0: grid.y carries the matrix, as before.
    uint32_t flat_tiles; uint64_t out_batch;
};

__device__ __forceinline__ float4 ldg4(const float *p, bool ok) {
    return ok ? wg_ld_u(p) : make_float4(0.f, 0.f, 0.f, 0.f); // (any element-aligned address: wg_internal.hpp)
}
// epilogue element: alpha * acc (+ beta * old). (alpha, beta) = (1, 0) returns acc unchanged and never reads old.
__device__ __forceinline__ void store_c(float *p, float4 v, float alpha, float beta) {
    if (alpha != 1.f) { v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha; }
    if (beta != 0.f) {
        const float4 c = wg_ld_u(p);
        v.x = fmaf(beta, c.x, v.x); v.y = fmaf(beta, c.y, v.y); v.z = fmaf(beta, c.z, v.z); v.w = fmaf(beta, c.w, v.w);
    }
    wg_st_u(p, v);
}
__device__ __forceinline__ float comp(const float4 &v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }

// workgroup id -> tile (bijective). Hardware deals ids round-robin to the 8 XCDs (guide T1); XCD x = id % 8 takes a contiguous range
// of the order index o, and o walks strips of 4 tile rows column by column: the workgroups an XCD runs at a time cover a compact patch
// of the output (4 A panels x a run of B panels through its L2), and the tiles sharing an A panel of a tall-skinny product run together.
#ifndef WG_F32_TILE_ORDER
#define WG_F32_TILE_ORDER 1 // 1 = strips of 4 tile rows; 0 = column-major (experiments)
#endif
__device__ __forceinline__ void tile_of(uint32_t bid, uint32_t tiles_m, uint32_t tiles_n, uint32_t &tm, uint32_t &tn) {
    const uint32_t nwg = tiles_m * tiles_n;
    const uint32_t q = nwg / 8u, r = nwg % 8u, xcd = bid % 8u, local = bid / 8u;
    const uint32_t o = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + local;
    if (!WG_F32_TILE_ORDER) { tm = o % tiles_m; tn = o / tiles_m; return; }
    const uint32_t strip = o / (4u * tiles_n);
    const uint32_t within = o - strip * 4u * tiles_n;
    const uint32_t h = min(4u, tiles_m - 4u * strip);
    tn = within / h;
    tm = 4u * strip + (within - tn * h);
}

// One k-tile (16 k) = 2 substeps of 8 k = 2 x 32 MFMAs per wave. The loop body is hand-placed in "slots" (one MFMA +
// at most one other memory op) pinned with sched_barrier(0); the compiler still inserts every s_waitcnt, so the
// dependencies stay correct by construction:
//   substep 0: MFMAs on fragment set 0 | slots 0-5: LDS reads of substep 1's fragments -> set 1
//                                      | slots 8-13: global loads of tile t+1 -> staging registers
//   substep 1: MFMAs on fragment set 1 | slots 0-5: staging registers -> LDS buffer (t+1)&1
//                                      | after slot 7: barrier (tile t+1 published; buffer t&1 may be refilled next round)
//                                      | slots 8-13: LDS reads of tile t+1's substep-0 fragments -> set 0
// so no LDS latency is exposed at a tile boundary and the barrier sits in the middle of a run of MFMAs.
// EDGE = false: interior tiles, no predication (no branches in the loop).
// B_NC (round 6, Gemm only): m2 is given contiguous along N -- element (k, n) at b + n + k * ldb -- which is what the ROW-major GemmTr is in column-major terms
// (api.hip wg_gemm_rm; shape.wgsl:49-57). Its tile is staged as Bs[16 k][128 n], the global layout, like Gemm's A; a lane reads the float2 of columns 2 i, 2 i + 1
// at one k, whose two components feed the two N-tiles of its wave: N-tile u holds columns {2 i + u}, undone in the epilogue's column index. The k of every MFMA is
// the one the k-contiguous form uses, so the result is bit-identical to transposing m2 first.
template <bool TRANS_A, bool EDGE, bool B_NC = false>
__device__ __forceinline__ void gemm_f32_tile(const GemmArgs &g, float *As, float *Bs, const float *A, const float *B, float *C,
                                              uint32_t m0, uint32_t n0) {
    static_assert(!(B_NC && TRANS_A), "n-contiguous m2: Gemm only");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;

    floatx16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

    wg_f4 ra[4], rb[2]; // staging registers of the next tile
    // Interior tiles issue the loads from inline asm: hipcc otherwise sinks them from their slot down to the LDS stores that
    // consume them and waits for each right there (measured: full memory latency exposed once per k-tile, -12 %). An asm load
    // is invisible to the compiler's vmcnt bookkeeping, so wait_staged() is the explicit wait in front of the first store.
    auto ldg = [&](wg_f4 &dst, const float *p, bool ok) {
        if constexpr (EDGE) {
            float4 v = ldg4(p, ok);
            dst = wg_f4{ v.x, v.y, v.z, v.w };
        } else {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));
        }
    };
    auto wait_staged = [&]() {
        if constexpr (!EDGE)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(rb[0]), "+v"(rb[1])::"memory");
    };
    auto load_one = [&](uint32_t k0, int r) { // r: 0..3 -> A, 4..5 -> B
        if (WG_ABLATE & 2) { if (r < 4) ra[r] = wg_f4{ 1, 2, 3, 4 }; else rb[r - 4] = wg_f4{ 1, 2, 3, 4 }; return; }
        if (r < 4) {
            const int f = tid + kThreads * r;
            if constexpr (!TRANS_A) { // A[m][k] at a + k*lda + m : float4 along m
                const uint32_t m = m0 + 4u * (f & 63), k = k0 + (f >> 6);
                ldg(ra[r], A + (uint64_t)k * g.lda + m, m < g.M && k < g.K);
            } else { // op(A)[m][k] at a + m*lda + k : float4 along k
                const uint32_t m = m0 + (f >> 2), k = k0 + 4u * (f & 3);
                ldg(ra[r], A + (uint64_t)m * g.lda + k, m < g.M && k < g.K);
            }
        } else if constexpr (B_NC) { // B[k][n] at b + k*ldb + n : float4 along n, 32 per k row
            const int f = tid + kThreads * (r - 4);
            const uint32_t n = n0 + 4u * (f & 31), k = k0 + (f >> 5);
            ldg(rb[r - 4], B + (uint64_t)k * g.ldb + n, n < g.N && k < g.K);
        } else {
            const int f = tid + kThreads * (r - 4);
            const uint32_t n = n0 + (f >> 2), k = k0 + 4u * (f & 3);
            ldg(rb[r - 4], B + (uint64_t)n * g.ldb + k, n < g.N && k < g.K);
        }
    };
    auto store_one = [&](int buf, int r) {
        if (WG_ABLATE & 2) return;
        if (r < 4) {
            const int f = tid + kThreads * r;
            if constexpr (!TRANS_A) {
                *reinterpret_cast<wg_f4 *>(&As[buf * A_TILE + (f >> 6) * BM + 4 * (f & 63)]) = ra[r];
            } else {
                const int mm = f >> 2, ch = f & 3;
                *reinterpret_cast<wg_f4 *>(&As[buf * A_TILE + mm * BK + 4 * (ch ^ ((mm >> 2) & 3))]) = ra[r];
            }
        } else if constexpr (B_NC) {
            const int f = tid + kThreads * (r - 4);
            *reinterpret_cast<wg_f4 *>(&Bs[buf * B_TILE + (f >> 5) * BN + 4 * (f & 31)]) = rb[r - 4];
        } else {
            const int f = tid + kThreads * (r - 4);
            const int nn = f >> 2, ch = f & 3;
            *reinterpret_cast<wg_f4 *>(&Bs[buf * B_TILE + nn * BK + 4 * (ch ^ ((nn >> 2) & 3))]) = rb[r - 4];
        }
    };
    float4 af[2][4], bf[2][2]; // fragment sets (by substep parity)
    auto read_one = [&](int buf, int ks, int r, int set) { // r: 0..3 -> A, 4..5 -> B
        const int chunk = 2 * ks + h; // this half-wave's 4 consecutive k within the 16-deep tile
        if (WG_ABLATE & 4) { if (r < 4) af[set][r] = make_float4(buf, ks, r, set); else bf[set][r - 4] = make_float4(buf, ks, r, set); return; }
        if (r < 4) {
            if constexpr (!TRANS_A) { // af[s]: rows 4i..4i+3 at k = 4*chunk + s
                af[set][r] = *reinterpret_cast<const float4 *>(&As[buf * A_TILE + (4 * chunk + r) * BM + wm * 128 + 4 * i]);
            } else { // af[t]: row 32t + i, k = 4*chunk .. +3
                const int mm = wm * 128 + 32 * r + i;
                af[set][r] = *reinterpret_cast<const float4 *>(&As[buf * A_TILE + mm * BK + 4 * (chunk ^ ((mm >> 2) & 3))]);
            }
        } else if constexpr (B_NC) { // read r - 4 takes k = 4 chunk + 2 (r - 4) and + 1: columns 2 i, 2 i + 1 -> component s of bf[0] / bf[1]
            const float *bk = &Bs[buf * B_TILE + (4 * chunk + 2 * (r - 4)) * BN + wn * 64 + 2 * i];
            const float2 b0 = *reinterpret_cast<const float2 *>(bk), b1 = *reinterpret_cast<const float2 *>(bk + BN);
            if (r == 4) { bf[set][0].x = b0.x; bf[set][1].x = b0.y; bf[set][0].y = b1.x; bf[set][1].y = b1.y; }
            else { bf[set][0].z = b0.x; bf[set][1].z = b0.y; bf[set][0].w = b1.x; bf[set][1].w = b1.y; }
        } else {
            const int nn = wn * 64 + 32 * (r - 4) + i;
            bf[set][r - 4] = *reinterpret_cast<const float4 *>(&Bs[buf * B_TILE + nn * BK + 4 * (chunk ^ ((nn >> 2) & 3))]);
        }
    };
    auto mfma_slot = [&](int set, int j) { // j: 0..31 -> (s, t, u); consecutive MFMAs hit different accumulators
        const int sidx = j >> 3, t = (j >> 1) & 3, u = j & 1;
        const float a = TRANS_A ? comp(af[set][t], sidx) : comp(af[set][sidx], t);
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, comp(bf[set][u], sidx), acc[t][u], 0, 0, 0);
    };

    const uint32_t nk = (g.K + BK - 1) / BK;
    auto tile_body = [&](uint32_t kt, auto has_next) {
        constexpr bool NEXT = decltype(has_next)::value;
        const int buf = kt & 1;
#pragma unroll
        for (int j = 0; j < 32; ++j) { // substep 0
            mfma_slot(0, j);
            if (j < 6) read_one(buf, 1, j, 1);
            if (NEXT && j >= 8 && j < 14) load_one((kt + 1) * BK, j - 8);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) { // substep 1
            mfma_slot(1, j);
            if (NEXT && j == 0) wait_staged(); // issued >= 18 MFMA slots (>= 1100 cycles) ago
            if (NEXT && j < 6) store_one(buf ^ 1, j);
            if (NEXT && j == 7) __syncthreads();
            if (NEXT && j >= 8 && j < 14) read_one(buf ^ 1, 0, j - 8, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (nk > 0) {
#pragma unroll
        for (int r = 0; r < 6; ++r) load_one(0, r);
        wait_staged();
#pragma unroll
        for (int r = 0; r < 6; ++r) store_one(0, r);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 6; ++r) read_one(0, 0, r, 0);
        __builtin_amdgcn_sched_barrier(0);
        for (uint32_t kt = 0; kt + 1 < nk; ++kt) tile_body(kt, std::true_type{});
        tile_body(nk - 1, std::false_type{});
    }

    // epilogue. C/D map of the 32x32 MFMA: lane l, register e -> row (e&3) + 8*(e>>2) + 4*(l>>5), col l&31.
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t col = n0 + wn * 64 + (B_NC ? 2 * i + u : 32 * u + i);
        if (EDGE && col >= g.N) continue;
        float *cc = C + (uint64_t)col * g.ldc;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) { // e >> 2
            if constexpr (!TRANS_A) {
                // M-tile t holds rows 4*row_mfma + t: (t, e&3) enumerate 16 consecutive rows from 32*gq + 16*h
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t row = m0 + wm * 128 + 32 * gq + 16 * h + 4 * q;
                    if (!EDGE || row < g.M)
                        store_c(cc + row, make_float4(acc[0][u][4 * gq + q], acc[1][u][4 * gq + q], acc[2][u][4 * gq + q], acc[3][u][4 * gq + q]),
                                g.alpha, g.beta);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t row = m0 + wm * 128 + 32 * t + 8 * gq + 4 * h;
                    if (!EDGE || row < g.M)
                        store_c(cc + row, make_float4(acc[t][u][4 * gq + 0], acc[t][u][4 * gq + 1], acc[t][u][4 * gq + 2], acc[t][u][4 * gq + 3]),
                                g.alpha, g.beta);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Interior tiles: no staging registers at all. Each k-tile (A 16 KiB + B 8 KiB) is brought in by LDS-DMA
// (global_load_lds_dwordx4: 16 B per lane, 1 KiB per wave-instruction) into a ring of 3 LDS slots, TWO tiles ahead of
// its use, and awaited with a counted `s_waitcnt vmcnt(6)` (this wave's 6 pieces of the newest tile may stay in flight).
// That is a full k-tile (~8000 cycles with two workgroups per CU) of lead time -- first-touch MALL/HBM latency was
// measured to exceed the ~2600 cycles a register-staged prefetch of the next tile can give within 256 VGPRs.
//   A (NN) slot image [16 k][256 m]  == the global layout: one DMA piece per k-row;
//   B / TN-A slot image [row][16 k] (64-byte rows) with the chunk swizzle c ^ ((row>>2)&3) applied to the SOURCE address.
// The DMA is issued from inline asm so that hipcc does not serialise it against the LDS reads of the other slots.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SLOT_FLOATS = A_TILE + B_TILE; // 6144 floats = 24 KiB
constexpr int NRING = 3;

__device__ __forceinline__ void dma16(uint32_t voff, const void *sbase, uint32_t lds_dst) {
    if (WG_ABLATE & 2) return;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst));
}

template <bool TRANS_A, bool B_NC = false>
__device__ __forceinline__ void gemm_f32_tile_dma(const GemmArgs &g, float *smem, const float *A, const float *B, float *C,
                                                  uint32_t m0, uint32_t n0) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;

    floatx16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

    // ---- DMA addressing: this wave stages 4 pieces of A and 2 of B per k-tile ----
    uint32_t a_voff[4], b_voff[2];
    const float *a_base, *b_base = B_NC ? B + n0 : B + (uint64_t)n0 * g.ldb;
    if constexpr (TRANS_A) a_base = A + (uint64_t)m0 * g.lda; else a_base = A + m0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if constexpr (!TRANS_A) { // piece = k-row 4*wave + q: 256 consecutive m, lane -> m = 4*lane
            a_voff[q] = ((4u * wave + q) * g.lda + 4u * lane) * 4u;
        } else { // piece P = 4*wave + q: rows 16P..16P+15, lane -> row 16P + (lane>>2), chunk (lane&3) ^ (lane>>4)
            const uint32_t row = 16u * (4u * wave + q) + (lane >> 2);
            a_voff[q] = (row * g.lda + 4u * ((lane & 3) ^ (lane >> 4))) * 4u;
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if constexpr (B_NC) { // piece P = 2*wave + q: k-rows 2P, 2P + 1 of 128 consecutive n, lane -> k row 2P + (lane>>5), n = 4*(lane&31): the slot image [16 k][128 n]
            b_voff[q] = ((2u * (2u * wave + q) + (lane >> 5)) * g.ldb + 4u * (lane & 31)) * 4u;
            continue;
        }
        const uint32_t row = 16u * (2u * wave + q) + (lane >> 2);
        b_voff[q] = (row * g.ldb + 4u * ((lane & 3) ^ (lane >> 4))) * 4u;
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    auto dma_piece = [&](uint32_t slot, uint32_t k0, int p) { // p: 0..3 -> A pieces, 4..5 -> B pieces
        const uint32_t sl = lds_base + slot * (SLOT_FLOATS * 4);
        if (p < 4) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(sl + (4 * wave + p) * 1024);
            if constexpr (TRANS_A) dma16(a_voff[p], a_base + k0, dst);
            else dma16(a_voff[p], a_base + (uint64_t)k0 * g.lda, dst);
        } else {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(sl + A_TILE * 4 + (2 * wave + (p - 4)) * 1024);
            if constexpr (B_NC) dma16(b_voff[p - 4], b_base + (uint64_t)k0 * g.ldb, dst);
            else dma16(b_voff[p - 4], b_base + k0, dst);
        }
    };

    float4 af[2][4], bf[2][2]; // fragment sets (by substep parity)
    auto read_one = [&](uint32_t slot, int ks, int r, int set) { // r: 0..3 -> A, 4..5 -> B
        const float *As = smem + slot * SLOT_FLOATS;
        const float *Bs = As + A_TILE;
        const int chunk = 2 * ks + h;
        if (WG_ABLATE & 4) { if (r < 4) af[set][r] = make_float4(slot, ks, r, set); else bf[set][r - 4] = make_float4(slot, ks, r, set); return; }
        if (r < 4) {
            if constexpr (!TRANS_A) {
                af[set][r] = *reinterpret_cast<const float4 *>(&As[(4 * chunk + r) * BM + wm * 128 + 4 * i]);
            } else {
                const int mm = wm * 128 + 32 * r + i;
                af[set][r] = *reinterpret_cast<const float4 *>(&As[mm * BK + 4 * (chunk ^ ((mm >> 2) & 3))]);
            }
        } else if constexpr (B_NC) { // (gemm_f32_tile's: k = 4 chunk + 2 (r - 4) and + 1, columns 2 i and 2 i + 1)
            const float *bk = &Bs[(4 * chunk + 2 * (r - 4)) * BN + wn * 64 + 2 * i];
            const float2 b0 = *reinterpret_cast<const float2 *>(bk), b1 = *reinterpret_cast<const float2 *>(bk + BN);
            if (r == 4) { bf[set][0].x = b0.x; bf[set][1].x = b0.y; bf[set][0].y = b1.x; bf[set][1].y = b1.y; }
            else { bf[set][0].z = b0.x; bf[set][1].z = b0.y; bf[set][0].w = b1.x; bf[set][1].w = b1.y; }
        } else {
            const int nn = wn * 64 + 32 * (r - 4) + i;
            bf[set][r - 4] = *reinterpret_cast<const float4 *>(&Bs[nn * BK + 4 * (chunk ^ ((nn >> 2) & 3))]);
        }
    };
    auto mfma_slot = [&](int set, int j) {
        const int sidx = j >> 3, t = (j >> 1) & 3, u = j & 1;
        const float a = TRANS_A ? comp(af[set][t], sidx) : comp(af[set][sidx], t);
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, comp(bf[set][u], sidx), acc[t][u], 0, 0, 0);
    };

    const uint32_t nk = g.K / BK; // whole k-tiles (>= 1); a K % 16 remainder is handled after the pipelined loop
    // one k-tile on ring slot `cur`; `nxt` holds tile t+1; the DMA of tile t+2 goes to `dst`
    auto tile_body = [&](uint32_t cur, uint32_t nxt, uint32_t dst, uint32_t k_dma, auto do_dma, auto has_next) {
#pragma unroll
        for (int j = 0; j < 32; ++j) { // substep 0
            mfma_slot(0, j);
            if (j < 6) read_one(cur, 1, j, 1);
            if constexpr (decltype(do_dma)::value) {
                if (j >= 8 && ((j - 8) & 3) == 0 && (j - 8) / 4 < 6) dma_piece(dst, k_dma, (j - 8) / 4); // slots 8,12,...,28
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) { // substep 1
            mfma_slot(1, j);
            if constexpr (decltype(has_next)::value) {
                if (j == 3) { // tile t+1 has landed (it was issued a whole tile ago); tile t+2's 6 pieces may stay in flight
                    if constexpr (decltype(do_dma)::value) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                if (j >= 4 && j < 10) read_one(nxt, 0, j - 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto next_slot = [](uint32_t x) { return x + 1 == NRING ? 0u : x + 1; };

    for (uint32_t t = 0; t < 2 && t < nk; ++t) {
#pragma unroll
        for (int p = 0; p < 6; ++p) dma_piece(t, t * BK, p);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 6; ++r) read_one(0, 0, r, 0);
    __builtin_amdgcn_sched_barrier(0);

    uint32_t cur = 0, t = 0;
    for (; t + 2 < nk; ++t) {
        const uint32_t nxt = next_slot(cur);
        tile_body(cur, nxt, next_slot(nxt), (t + 2) * BK, std::true_type{}, std::true_type{});
        cur = nxt;
    }
    for (; t + 1 < nk; ++t) {
        const uint32_t nxt = next_slot(cur);
        tile_body(cur, nxt, 0, 0, std::false_type{}, std::true_type{});
        cur = nxt;
    }
    tile_body(cur, cur, 0, 0, std::false_type{}, std::false_type{});

    // K % 16 != 0 (a multiple of 4 by the operator's precondition): one more k-tile, staged through registers into slot 0 with the
    // missing k zero-filled (same slot image as the DMA writes), multiplied like any other. Once per tile: not worth pipelining; it
    // keeps a ragged K on this path instead of the predicated one (4096 x 4096 x 4100: 119 TFLOP/s there).
    if (nk * BK < g.K) {
        __syncthreads(); // every wave is done with the ring
        const uint32_t k0 = nk * BK;
        float *As = smem, *Bs = smem + A_TILE;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = tid + kThreads * r;
            if constexpr (!TRANS_A) {
                const uint32_t k = k0 + (f >> 6);
                const float4 v = ldg4(A + (uint64_t)k * g.lda + m0 + 4u * (f & 63), k < g.K);
                *reinterpret_cast<float4 *>(&As[(f >> 6) * BM + 4 * (f & 63)]) = v;
            } else {
                const int mm = f >> 2, ch = f & 3;
                const uint32_t k = k0 + 4u * ch;
                const float4 v = ldg4(A + (uint64_t)(m0 + mm) * g.lda + k, k < g.K);
                *reinterpret_cast<float4 *>(&As[mm * BK + 4 * (ch ^ ((mm >> 2) & 3))]) = v;
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int f = tid + kThreads * r;
            if constexpr (B_NC) {
                const uint32_t k = k0 + (f >> 5);
                const float4 v = ldg4(B + (uint64_t)k * g.ldb + n0 + 4u * (f & 31), k < g.K);
                *reinterpret_cast<float4 *>(&Bs[(f >> 5) * BN + 4 * (f & 31)]) = v;
                continue;
            }
            const int nn = f >> 2, ch = f & 3;
            const uint32_t k = k0 + 4u * ch;
            const float4 v = ldg4(B + (uint64_t)(n0 + nn) * g.ldb + k, k < g.K);
            *reinterpret_cast<float4 *>(&Bs[nn * BK + 4 * (ch ^ ((nn >> 2) & 3))]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 6; ++r) read_one(0, 0, r, 0);
#pragma unroll
        for (int r = 0; r < 6; ++r) read_one(0, 1, r, 1);
#pragma unroll
        for (int j = 0; j < 32; ++j) mfma_slot(0, j);
#pragma unroll
        for (int j = 0; j < 32; ++j) mfma_slot(1, j);
    }

    // epilogue (interior: no predication)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float *cc = C + (uint64_t)(n0 + wn * 64 + (B_NC ? 2 * i + u : 32 * u + i)) * g.ldc;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            if constexpr (!TRANS_A) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    store_c(cc + m0 + wm * 128 + 32 * gq + 16 * h + 4 * q,
                            make_float4(acc[0][u][4 * gq + q], acc[1][u][4 * gq + q], acc[2][u][4 * gq + q], acc[3][u][4 * gq + q]), g.alpha, g.beta);
            } else {
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4)
                    store_c(cc + m0 + wm * 128 + 32 * t4 + 8 * gq + 4 * h,
                            make_float4(acc[t4][u][4 * gq + 0], acc[t4][u][4 * gq + 1], acc[t4][u][4 * gq + 2], acc[t4][u][4 * gq + 3]), g.alpha, g.beta);
            }
        }
    }
}

template <bool TRANS_A, bool B_NC = false>
__global__ __launch_bounds__(kThreads, 2) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[NRING * SLOT_FLOATS]; // 72 KiB: 3 DMA slots, or 2 staged buffers on edge tiles
    uint32_t tm, tn, id = blockIdx.x + g.tile_base, z, split;
    if (g.flat_tiles) { z = id / g.flat_tiles; id -= z * g.flat_tiles; split = blockIdx.y; } // (workgroup-uniform)
    else { z = blockIdx.y / g.nsplit; split = blockIdx.y % g.nsplit; }
    tile_of(id, g.tiles_m, g.tiles_n, tm, tn);
    const uint32_t m0 = tm * BM, n0 = tn * BN;
    const uint32_t k_begin = split * g.k_per_split; // split-K: this workgroup's K range (k_per_split is a multiple of BK)
    GemmArgs gl = g;
    gl.K = min(g.K - k_begin, g.k_per_split);
    const float *A = g.a + z * g.a_batch + (TRANS_A ? (uint64_t)k_begin : (uint64_t)k_begin * g.lda);
    const float *B = g.b + z * g.b_batch + (B_NC ? (uint64_t)k_begin * g.ldb : (uint64_t)k_begin);
    float *C = g.c + ((uint64_t)z * (g.flat_tiles ? 1u : g.nsplit) + (g.flat_tiles ? 0u : split)) * g.c_batch;
    if (g.tail_tiles > 0) { // dense partial tile: C[n * BM + m] with (m, n) relative to the tile
        C = g.c + ((uint64_t)split * g.tail_tiles + blockIdx.x) * (uint64_t)(BM * BN) - ((uint64_t)n0 * BM + m0);
        gl.ldc = BM;
    }
    // workgroup-uniform: the whole tile is inside the matrices, at least one whole k-tile, and 32-bit DMA offsets suffice
    const bool interior = (m0 + BM <= gl.M) && (n0 + BN <= gl.N) && gl.K >= (uint32_t)BK && gl.dma_ok;
    if (interior) gemm_f32_tile_dma<TRANS_A, B_NC>(gl, smem, A, B, C, m0, n0);
    else gemm_f32_tile<TRANS_A, true, B_NC>(gl, smem, smem + 2 * A_TILE, A, B, C, m0, n0);
}

// Tail split (tile quantisation): see wgk_gemm_f32. Adds a tail tile's partials in ASCENDING split order (deterministic) and writes
// it with the usual alpha / beta / edge rules. grid = (tail tiles, BN / 4): 4 columns per block, float4 (4 rows) per thread.
__global__ __launch_bounds__(256) void gemm_f32_tail_reduce(GemmArgs g) {
    uint32_t tm, tn, id = blockIdx.x + g.tile_base, z = 0;
    if (g.flat_tiles) { z = id / g.flat_tiles; id -= z * g.flat_tiles; }
    tile_of(id, g.tiles_m, g.tiles_n, tm, tn);
    const uint32_t rl = 4u * (threadIdx.x & 63u), cl = blockIdx.y * 4u + (threadIdx.x >> 6);
    const uint32_t row = tm * BM + rl, col = tn * BN + cl;
    if (row >= g.M || col >= g.N) return; // M % 4 == 0: the 4 rows are all in or all out
    const float *p = g.c + (uint64_t)blockIdx.x * (uint64_t)(BM * BN) + (uint64_t)cl * BM + rl;
    float4 s = *reinterpret_cast<const float4 *>(p);
    for (uint32_t i = 1; i < g.nsplit; ++i) {
        const float4 q = *reinterpret_cast<const float4 *>(p + (uint64_t)i * g.tail_tiles * (uint64_t)(BM * BN));
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    float *o = g.out_c + (uint64_t)z * g.out_batch + (uint64_t)col * g.out_ldc + row;
    if (g.out_alpha != 1.f) { s.x *= g.out_alpha; s.y *= g.out_alpha; s.z *= g.out_alpha; s.w *= g.out_alpha; }
    if (g.out_beta != 0.f) {
        const float4 c = wg_ld_u(o);
        s.x = fmaf(g.out_beta, c.x, s.x); s.y = fmaf(g.out_beta, c.y, s.y); s.z = fmaf(g.out_beta, c.z, s.z); s.w = fmaf(g.out_beta, c.w, s.w);
    }
    wg_st_u(o, s);
}

} // namespace

int wgk_gemm_f32(wg_ctx *ctx, bool trans, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats,
                 float *out, uint32_t out_ld, uint64_t out_batch, wgk_mat m1, wgk_mat m2, float alpha, float beta) {
    if (M == 0 || N == 0 || nmats == 0) return WG_OK;
    if (nmats > 65535) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: more than 65535 matrices in one call");
    // Few output ROWS (M <= 64, many columns): every tiling here is built around tall row blocks, so the product is computed transposed,
    // C^T (N x M) = op(B)^T op(A)^T, which is a GemmTr with few columns on m1' = m2 (K x N, already k-contiguous) and m2' = op(A)^T as
    // a K x M column-major matrix -- m1 itself for GemmTr, a transposed copy of the tiny m1 for Gemm -- followed by a transpose of the
    // small result. 16 x 4096 x 4096: 79 us on the 256 x 128 tiles, 34 us this way. beta needs the old output inside the product: not taken then.
    // 65 .. 128 rows: a 256 x 128 tile would be at most half full; transposed, the product has <= 128 COLUMNS -- one full-width tile column of
    // 256-row tiles -- at the price of a transposed copy of the small m1 (Gemm only) and a transpose of the small result: 128 x 11008 x 4096
    // 249 -> 137 us, 128 x 14336 x 4096 305 -> 174 (vendor 117 on the first; profiles/r03_evidence.md section 11).
    // (N <= 4096 has the 64-column panels below: 128 x 4096 x 4096 49 us there, 68 this way)
    const bool mid_forced = ctx->tuning[WG_TUNE_F32_MID] > 1; // (tests / sweeps: a forced tile of the mid family goes past the few-row / few-column paths)
    // Few 64 x 64 tiles with a long K (64 x 4096 x 4096, 256 x 256 x 32768): the mid family's k-split tile with K cut across workgroups -- f32 slabs +
    // the ordered reduce, but on tiles small enough that the slabs are small. Model (tools/f32_mid_sweep.py with SPLITS=..., profiles/r04_f32_split_sweep.txt):
    // ceil(tiles x ns / CUs) rounds of a K / ns tile (+16 % alone on a CU, the co-resident curve beyond) + 1.5 us per tile + 3 us, slabs written at 3.5 TB/s,
    // reduced at 7 TB/s + 4 us. Measured: 64 x 4096 x 4096 29 -> 26 us (vendor 25), 64 x 11008 x 4096 76 -> 57 (61), 256 x 256 x 32768 140 -> 43 (139).
    const int cus0 = ctx->compute_units > 0 ? ctx->compute_units : 256;
    auto mid_split_plan = [&](double &est_out) -> uint32_t {
        const uint64_t t64 = (uint64_t)((M + 63u) / 64u) * ((N + 63u) / 64u) * nmats;
        uint32_t best_ns = 1;
        est_out = 1e30;
        if (K < 1024u || t64 > (uint64_t)cus0) return 1; // (up to one 64 x 64 tile per CU: 64 x 11008 x 4096 -- 172 tiles -- 76 -> 57 us with 4 splits, vendor 61)
        const double ob = (double)M * N * nmats * 4.0;
        static const uint32_t opts[] = { 2, 3, 4, 6, 8, 12, 16, 24, 32 };
        for (uint32_t ns : opts) {
            if (K / ns < 256u || (uint64_t)nmats * ns > 65535u) break;
            const double r = (double)((t64 * ns + (uint64_t)cus0 - 1) / (uint64_t)cus0);
            const double loop = r <= 1.0 ? 1.16 : 1.03 + 0.02 * log2(r);
            const double kps = (double)((((K + 31u) / 32u + ns - 1u) / ns) * 32u);
            const double est = r * (2.0 * 64 * 64 * kps / 614400.0 * loop + 1.5) + 3.0 + ns * ob / 3.5e6 + 4.0 + ns * ob / 7.0e6;
            if (est < est_out) { est_out = est; best_ns = ns; }
        }
        return best_ns;
    };
    // (a forced few-column kernel -- WG_TUNE_F32_SKINNY / _PANELS = 1, "whenever applicable" -- goes past this early path to the kernel it names)
    const bool other_forced = ctx->tuning[WG_TUNE_F32_SKINNY] == 1 || ctx->tuning[WG_TUNE_F32_PANELS] == 1;
    if (ctx->tuning[WG_TUNE_F32_MID] != 0 && !mid_forced && !other_forced && (M <= 64 || N <= 64) && M >= 48 && N >= 48 &&
        wgk_gemm_f32_mid_ok(M, N, K, nmats, m1, m2)) {
        double est;
        uint32_t ns = mid_split_plan(est);
        // ... and the same tile UNSPLIT when the output is about one to two 64 x 64 tiles per CU (the few-row / few-column paths below stream the long operand
        // through wave-private rings and leave the matrix cores at ~45 %): 64 x 16384 x 1024 31 -> 22 us (vendor 21.7), 64 x 16384 x 512 22 -> 13.5 (19),
        // 64 x 14336 x 4096 73 -> 64 (62), 64 x 32768 x 1024 43 -> 38 (35); the same model, without slabs and reduce.
        const uint64_t t64 = (uint64_t)((M + 63u) / 64u) * ((N + 63u) / 64u) * nmats;
        if (t64 * 4u >= 3ull * (uint64_t)cus0 && t64 <= 2ull * (uint64_t)cus0 && K >= 128u) {
            const double r = (double)((t64 + (uint64_t)cus0 - 1) / (uint64_t)cus0);
            const double est1 = r * (2.0 * 64 * 64 * (double)K / 614400.0 * (r <= 1.0 ? 1.16 : 1.03 + 0.02 * log2(r)) + 1.5) + 3.0;
            if (est1 < est) { est = est1; ns = 1; }
        }
        if (est < 1e29) return wgk_gemm_f32_mid(ctx, trans, 64, 64, M, N, K, nmats, out, out_ld, out_batch, m1, m2, alpha, beta, ns);
    }
    if (!mid_forced && M > 64 && M <= 128 && N > 4096 && N % 4u == 0 && K >= 128 && beta == 0.f) { // (N % 4: it is the transposed product's row count)
        const uint64_t at_elems = trans ? 0 : (uint64_t)K * M, ct_elems = (uint64_t)N * M;
        void *ws = nullptr;
        if (int rc = wg_ctx_pad_workspace(ctx, (size_t)((at_elems + ct_elems) * nmats * sizeof(float)), &ws)) return rc;
        float *at = (float *)ws, *ct = at + at_elems * nmats;
        wgk_mat a2 = m1; // K x M, column m contiguous in k
        if (!trans) {
            if (int rc = wgk_transpose(ctx, WG_F32, M, K, nmats, m1.ptr, m1.ld, m1.batch, at, K, at_elems)) return rc;
            a2 = wgk_mat{ at, K, at_elems };
        }
        if (int rc = wgk_gemm_f32(ctx, true, N, M, K, nmats, ct, N, ct_elems, m2, a2, alpha, 0.f)) return rc;
        return wgk_transpose(ctx, WG_F32, N, M, nmats, ct, N, ct_elems, out, out_ld, out_batch);
    }
    if (!mid_forced && M <= 64 && N >= 512 && N % 4u == 0 && K >= 128 && beta == 0.f) {
        // the few-column GemmTr kernel takes m2' = op(A)^T either k-contiguous (GemmTr: m1 as it is) or with its columns contiguous (Gemm:
        // m1 as it is, "k-major"), and writes -- or its split-K reduce does -- straight into the transposed position: "row" n of C^T is
        // column n of C (out_ld apart), "column" m is row m (adjacent). No copy of anything.
        const bool dma_ok = (uint64_t)m2.ld * 32u * 4u < (1ull << 31) && (uint64_t)m1.ld * 64u * 4u < (1ull << 31);
        if (dma_ok) return wgk_gemm_f32_skinny(ctx, true, N, M, K, nmats, out, 1u, out_batch, m2, m1, alpha, 0.f, out_ld, /*m2_kmajor=*/!trans);
        // (leading dimensions beyond the kernel's 32-bit offsets: transposed copies and the general path)
        const uint64_t at_elems = trans ? 0 : (uint64_t)K * M, ct_elems = (uint64_t)N * M;
        void *ws = nullptr;
        if (int rc = wg_ctx_pad_workspace(ctx, (size_t)((at_elems + ct_elems) * nmats * sizeof(float)), &ws)) return rc;
        float *at = (float *)ws, *ct = at + at_elems * nmats;
        wgk_mat a2 = m1; // K x M, column m contiguous in k
        if (!trans) {
            if (int rc = wgk_transpose(ctx, WG_F32, M, K, nmats, m1.ptr, m1.ld, m1.batch, at, K, at_elems)) return rc;
            a2 = wgk_mat{ at, K, at_elems };
        }
        if (int rc = wgk_gemm_f32(ctx, true, N, M, K, nmats, ct, N, ct_elems, m2, a2, alpha, 0.f)) return rc;
        return wgk_transpose(ctx, WG_F32, N, M, nmats, ct, N, ct_elems, out, out_ld, out_batch);
    }
    // few output columns (a matrix applied to a handful of vectors): HBM-bound on A, see gemm_f32_skinny.hip. wg_ctx_set_tuning(WG_TUNE_F32_SKINNY, 0) disables
    // it (experiments / tests of the tiled kernel on these shapes).
    // (32-bit DMA offsets within a 32-row / 32-k block of m1 and within the 64 columns of m2: both variants build them)
    if (!mid_forced && N <= 64 && M >= 512 && K >= 128 && (uint64_t)m1.ld * 32u * 4u < (1ull << 31) && (uint64_t)m2.ld * 64u * 4u < (1ull << 31)) {
        if (ctx->tuning[WG_TUNE_F32_SKINNY] != 0) return wgk_gemm_f32_skinny(ctx, trans, M, N, K, nmats, out, out_ld, out_batch, m1, m2, alpha, beta);
    }
    GemmArgs g;
    g.a = (const float *)m1.ptr; g.lda = m1.ld; g.a_batch = m1.batch;
    g.b = (const float *)m2.ptr; g.ldb = m2.ld; g.b_batch = m2.batch;
    g.c = out; g.ldc = out_ld; g.c_batch = out_batch;
    g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta;
    g.tile_base = 0; g.tail_tiles = 0; g.out_c = out; g.out_ldc = out_ld; g.out_alpha = alpha; g.out_beta = beta;
    g.flat_tiles = 0; g.out_batch = out_batch;
    g.tiles_m = (M + BM - 1) / BM;
    g.tiles_n = (N + BN - 1) / BN;
    g.dma_ok = ((uint64_t)m1.ld * 256u * 4u < (1ull << 31)) && ((uint64_t)m2.ld * 128u * 4u < (1ull << 31)) ? 1u : 0u;
    const uint64_t tiles = (uint64_t)g.tiles_m * g.tiles_n;
    if (tiles > 0x7fffffffull) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: too many tiles");
    // Launch plan (tile quantisation). Measured on MI355X (profiles/r01_evidence.md section 12): a CU works through the workgroups it
    // is dealt at ~0.115 us per k of a 256 x 128 tile whether one or two of them are resident, so a launch takes
    //     W * (K / ns) * 0.115 us,  W = ceil(workgroups / CUs),
    // and cutting K into ns splits adds the f32 partial slabs (written at ~3.5 TB/s, + 3 us) and the ordered reduce (4 us + slabs read
    // at ~7 TB/s). Candidates: plain split-K with ns = 1 .. 16 (>= 128 k per split), and the "tail split" -- full rounds of one tile
    // per CU as they are, the r < CUs/2 tiles left over cut along K over the idle CUs (partials of those r tiles only).
    const int cus = ctx->compute_units > 0 ? ctx->compute_units : 256;
    const uint32_t ktiles = (K + BK - 1) / BK;
    const double us_per_k = 0.115, out_bytes = (double)M * N * nmats * 4.0;
    auto rounds = [&](uint64_t wgs) { return (double)((wgs + (uint64_t)cus - 1) / (uint64_t)cus); };
    uint32_t nsplit = 1;
    double best = rounds(tiles * nmats) * K * us_per_k;
    for (uint32_t ns = 2; ns <= 16 && ktiles / ns >= 8; ++ns) {
        if ((double)ns * out_bytes > (double)(512ull << 20)) break;
        const uint32_t kps = ((ktiles + ns - 1) / ns) * BK;
        const double t = rounds(tiles * nmats * ns) * kps * us_per_k + (ns * out_bytes / 3.5e6 + 3.0) + (4.0 + ns * out_bytes / 7.0e6);
        if (t < best * 0.97) { best = t; nsplit = ns; } // a split must pay for itself by a margin
    }
    uint32_t tail_r = 0, tail_sp = 1;
#ifndef WG_F32_TAIL_SPLIT
#define WG_F32_TAIL_SPLIT 1
#endif
#ifndef WG_F32_FLAT_BATCH
#define WG_F32_FLAT_BATCH 1 // 0: batches keep grid.y = matrix and never get the tail split (round 5; A/B builds)
#endif
    // More than one full wave of resident workgroups (2 per CU): the launch runs in waves of 2 x CUs tiles, K x 0.23 us each -- and the LAST
    // wave costs that much however few tiles it holds: completion times have drifted apart by then, a CU that finishes its pair is handed two
    // new workgroups at once, and the leftover r tiles end up two to a CU on r / 2 CUs (measured, K = 4096: 1024 tiles 1931 us, 1280 tiles
    // 2801, 1536 tiles 2824 -- five tiles per CU cost six; profiles/r03_evidence.md section 11). So the r = tiles mod (2 x CUs) leftover tiles are
    // cut along K into sp parts that run as their own launch (spread one per CU up to CUs workgroups: 0.136 us per k then, 0.23 per wave of
    // 2 x CUs beyond), with the split count that minimises wave time + the partial tiles' write and ordered reduce.
    bool tail_done = false;
    // (round 6: a batch is planned as a whole -- `all` = every matrix's tiles; its launches then number the tiles through the batch, GemmArgs::flat_tiles. Before, only a
    // single matrix got the tail split and a batch paid for its last, nearly empty wave: 4096 x 7168 x 2048 x 3 matrices 131 TFLOP/s against 143 for one)
    const uint64_t all = tiles * nmats;
    const bool flat_ok = all <= 0x7fffffffull && (nmats == 1 || WG_F32_FLAT_BATCH);
    if (WG_F32_TAIL_SPLIT && flat_ok && all > 2ull * (uint64_t)cus) {
        const uint64_t cap = 2ull * (uint64_t)cus;
        const uint32_t r = (uint32_t)(all % cap);
        if (r > 0 && nsplit == 1) {
            const double pair = 2.0 * us_per_k, lone = 0.136;
            double best_t = (double)K * pair * 0.97; // the leftover wave as it is (a split must pay for itself by a margin)
            uint32_t best_sp = 1;
            for (uint32_t sp = 2; sp <= 8 && ktiles / sp >= 8; ++sp) {
                const double part_bytes = (double)sp * r * BM * BN * 4.0;
                if (part_bytes > (double)(512ull << 20)) break;
                const uint32_t kps = ((ktiles + sp - 1) / sp) * BK;
                const uint64_t w = (uint64_t)r * sp, full = w / cap, rem = w % cap;
                const double t = (double)full * kps * pair + (rem == 0 ? 0.0 : rem <= (uint64_t)cus ? kps * lone : kps * pair) +
                                 (part_bytes / 3.5e6 + 3.0) + (4.0 + part_bytes / 7.0e6);
                if (t < best_t) { best_t = t; best_sp = sp; }
            }
            if (best_sp > 1) { tail_r = r; tail_sp = best_sp; }
            tail_done = true;
            // what this plan takes: the full waves of 2 x CUs tiles, then the leftover wave -- as it is, or cut along K
            best = (double)(all / cap) * K * pair + (best_sp > 1 ? best_t : (double)K * pair);
        }
    }
    if (WG_F32_TAIL_SPLIT && flat_ok && all > (uint64_t)cus && !tail_done) {
        const uint32_t r = (uint32_t)(all % (uint64_t)cus);
        uint32_t sp = r ? (uint32_t)cus / r : 0;
        if (sp > ktiles / 8u) sp = ktiles / 8u; // >= 8 k-tiles (128 k) per split
        if (r > 0 && r * 2u <= (uint32_t)cus && sp >= 2 && (size_t)sp * r * BM * BN * sizeof(float) <= (512ull << 20)) {
            const uint32_t kps = ((ktiles + sp - 1) / sp) * BK;
            const double part_bytes = (double)sp * r * BM * BN * 4.0;
            const double t = (double)((all - r) / (uint64_t)cus) * K * us_per_k + rounds((uint64_t)r * sp) * kps * us_per_k +
                             (part_bytes / 3.5e6 + 3.0) + (4.0 + part_bytes / 7.0e6);
            if (t < best) { best = t; nsplit = 1; tail_r = r; tail_sp = sp; }
        }
    }
    // Mid-size outputs: the small-tile family (gemm_f32_mid.hip) -- more tiles instead of a K cut or a detour over the few-column kernel.
    // Model, calibrated on tools/f32_mid_sweep.py (profiles/r04_f32_mid_sweep.txt): tiles are dealt round-robin, the busiest CU works through
    // ceil(tiles / CUs) of them, each at the matrix cores' rate (157.3 TFLOP/s / 256 per CU) less a loop cost by tile (barriers, issue, what
    // co-resident workgroups do not cover) plus ~1-1.5 us of its own (first fills, last stores, the k-split tiles' reduction); ~3 us per launch.
    //   2 x 2-wave tiles: 128 x 64 (+12 %), 64 x 128 (+20 %), 128 x 128 (+17 %, only while every CU gets at most one: three of them sharing a CU
    //   measured far worse, 4096^3 1409 us against 969 on 128 x 64);
    //   k-split tiles: 64 x 64 (GemmTr with a power-of-two leading dimension >= 1024 and several tiles per CU +30 %: 2048^3 148 us against
    //   124 for Gemm -- every row segment of a tile then comes from the same few memory channels), 96 x 96 / 96 x 64 / 64 x 96 (sizes that
    //   are multiples of 96: 1536^3 is 256 tiles of 96 x 96 -- 58 us against 70 on 64 x 64 and the vendor's 62), 64 x 32 / 32 x 64; with more
    //   than one round only from K = 256 up (their per-tile reduction does not amortise over a handful of k-tiles: 128^3 x 256 matrices 21 us against 16).
    double mid_est = 1e30;
    int mid_bm = 0, mid_bn = 0;
    uint32_t mid_ns = 1;
    const int mid_knob = ctx->tuning[WG_TUNE_F32_MID];
    // Mid-size means mid-size: from ~4 tiles of 256 x 128 per CU on, this file's kernel runs at 93-96 % of the matrix cores' rate and the small
    // tiles' extra barriers and fills only cost (8192^3: 7287 us here, 7980 on 64 x 64 tiles).
    // Short K on a large output (round 5, tools/f32_mid_sweep.py): this file's 256 x 128 tile pays its prologue and its 128 KiB store burst once per 128 .. 512 k, and the
    // 128 x 64 tile (two to four workgroups per CU, covering each other) is ahead at any output size -- plan -> 128 x 64 | vendor, us: Gemm 4096^2 x 128 50.5 -> 40.3 | 44.1,
    // x 256 78.7 -> 68.0 | 71.3, x 384 106.8 -> 99.6 | 98.9, x 512 135.2 -> 128.3 | 126.3 (x 768: 192 / 189, x 1024: 249 / 248: nothing left); 8192^2 x 128 167.5 -> 148.3 |
    // 156.6, 6144^2 x 256 185.4 -> 155.3 | 156.6, 2048^2 x 128 x 8 matrices 95.6 -> 77.3 | 80.6, 1024^2 x 64 x 64 124.0 -> 97.6 | 107.8; GemmTr (whose plan is the better
    // one at short K) 4096^2 x 128 43.3 -> 39.5, x 256 70.7 -> 67.6, x 384 100.3 -> 99.5, 6144^2 x 256 176.9 -> 155.7, 8192^2 x 256 278.0 -> 273.7, x 512 128.4 -> 130.2 (not taken).
    const bool short_k = (K <= 256u || (!trans && K <= 512u)) && 2u * tiles * nmats >= (uint64_t)cus; // (from half a tile per CU on: 1536 x 5120 x 384, 240 tiles, 57.8 -> 51.0 | 50.9)
    int sk_bm = 0, sk_bn = 0; // the better of 128 x 64 / 64 x 128 by the model below, when short_k
    double sk_est = 1e30;
    if (mid_knob != 0 && (mid_knob >= 1 || tiles * nmats <= 4ull * (uint64_t)cus || short_k) && wgk_gemm_f32_mid_ok(M, N, K, nmats, m1, m2)) {
        // (round 5: not only powers of two -- any multiple of 1024 floats, e.g. K = 3072: 768 x 5120 x 3072 GemmTr 217 us on 64 x 64 tiles against 181-187 on the others --
        // and Gemm with its rows a large power of two apart, whose k-steps then hit the same channels: 16384 x 256 x 3072, lda 16384, 216 us against 186-189)
        const bool pow2_ld = trans ? (m1.ld >= 1024u && m1.ld % 1024u == 0) : (m1.ld >= 8192u && (m1.ld & (m1.ld - 1u)) == 0);
        const bool pow2_ldb = m2.ld >= 8192u && (m2.ld & (m2.ld - 1u)) == 0; // rows of m2 a large power of two apart: the small tiles' row segments share few channels
        // { bm, bn, k-split family, loop cost in per cent (2 x 2-wave tiles; k-split tiles: on top of the curve below), tenths of a us per tile }
        static const int cand[9][5] = { { 128, 64, 0, 12, 10 }, { 64, 128, 0, 20, 10 }, { 64, 64, 1, 0, 15 }, { 96, 96, 1, 3, 20 }, { 96, 64, 1, 3, 18 }, { 64, 96, 1, 3, 18 },
                                        { 64, 32, 1, 4, 27 }, { 32, 64, 1, 4, 27 }, { 128, 128, 0, 17, 10 } };
        for (const auto &c : cand) {
            if (mid_knob > 1 && mid_knob != c[0] * 1000 + c[1]) continue;
            const uint64_t t = (uint64_t)((M + c[0] - 1) / c[0]) * ((N + c[1] - 1) / c[1]) * nmats;
            const double r = rounds(t);
            if (mid_knob <= 1) {
                if (c[0] == 128 && c[1] == 128 && r > 1.0) continue;
                if (c[2] && r > 1.0 && K < 256) continue;
            }
            // k-split tiles: one workgroup alone on a CU leaves its barriers uncovered (+16 %: 1024^3, 1536^3); co-resident ones cover each other
            // (+5 % at 4 rounds) until the small tiles' traffic shows (+9 % at 16 rounds, +13 % at 64: 2048^3 / 4096^3 / 8192^3 on 64 x 64)
            double loop = c[2] ? (r <= 1.0 ? 1.16 : 1.03 + 0.02 * log2(r)) + 0.01 * c[3] : 1.0 + 0.01 * c[3];
            if (c[0] == 64 && c[1] == 64 && pow2_ld && r > 2.0) loop = 1.30; // (three rounds and more: at two the tile is the best one -- 2048 x 1024 x 5120 GemmTr 146 us against 153-169, 8192 x 256 x 4096 122 against 124-137)
            // a 64 x 32 / 32 x 64 tile alone on its CU has 8 MFMAs per wave between two barriers: with a long K that shows (256 x 256 x 4096 x 8 matrices 44.7 us measured
            // against 38.5 by the curve above; 128 x 128 x 4096 x 32 matrices 62 -- there the K cut on 64 x 64 tiles is the better plan, 42)
            if (c[2] && r <= 1.0 && (c[0] == 32 || c[1] == 32) && K >= 2048u) loop += 0.25;
            if (c[2] && pow2_ldb) loop += 0.10; // 1024 x 1024 x 32768: 536 us on 64 x 32 tiles against 486 for this file's split-K plan
            // (round 6, tools/archive/r06/f32_ld_pad_probe.py: the 64 x 64 tile at ONE round of four workgroups per CU -- 2048^2, 1024 x 4096 outputs -- loses to 128 x 64 whenever
            // its rows are not fed ideally: GemmTr at any leading dimension (2048^3 172 us against 152; only ld % 1024 == 0 was penalised above, so ld = 2056 took the slow tile),
            // Gemm with rows of m1 or m2 that are not 64-byte aligned (165-178 against 153; K = 4096: 334 against 299). At 8-9 workgroups per CU neither shows: 3072^2 x 1024 stays on it.)
            if (c[0] == 64 && c[1] == 64 && 2u * t > 7ull * (uint64_t)cus && t <= 4ull * (uint64_t)cus && (trans || m1.ld % 16u != 0 || m2.ld % 16u != 0 || K >= 4096u)) loop += 0.15; // (K = 4096, everything aligned: 313 against 297)
            const double tile_us = 2.0 * c[0] * c[1] * (double)K / 614400.0; // one tile at a CU's full rate
            const double est = r * (tile_us * loop + 0.1 * c[4]) + 3.0;
            if (short_k && !c[2] && !(c[0] == 128 && c[1] == 128) && est < sk_est) { sk_est = est; sk_bm = c[0]; sk_bn = c[1]; }
            if (mid_knob <= 1 && tiles * nmats > 4ull * (uint64_t)cus) continue; // (past mid-size the family is a candidate for short K only)
            if (est < mid_est) { mid_est = est; mid_bm = c[0]; mid_bn = c[1]; mid_ns = 1; }
        }
        if (mid_knob <= 1 && tiles * nmats <= 4ull * (uint64_t)cus) { // few tiles, long K: the 64 x 64 tile with K cut across workgroups (mid_split_plan above)
            double est;
            const uint32_t ns = mid_split_plan(est);
            if (ns > 1 && est < mid_est) { mid_est = est; mid_bm = 64; mid_bn = 64; mid_ns = ns; }
        }
    }
    if (sk_bm && mid_knob < 1 && ctx->tuning[WG_TUNE_F32_PANELS] != 1 && !other_forced)
        return wgk_gemm_f32_mid(ctx, trans, sk_bm, sk_bn, M, N, K, nmats, out, out_ld, out_batch, m1, m2, alpha, beta, 1u);
    // Small outputs (few 256 x 128 tiles): 64-column panels of the few-column kernel (gemm_f32_skinny.hip) give 128 x 64 "tiles", eight
    // times as many, each streaming its rows through a wave-private ring at ~1.4 us per 32 k (+ ~2.5 us of pipeline fill per workgroup):
    // 1024^3 25 + 5 us instead of 33 + 7.
    // Batches of small matrices as well (M >= 32, N >= 16: a 256 x 128 tile is mostly empty there -- 64^3 x 1024 matrices 43 -> 25 us, 32^3 x 4096 84 -> 40).
    if (((N > 64 && M >= 128 && K >= 128) || (nmats > 1 && N >= 16 && M >= 32 && K >= 32)) && N <= 4096 && (uint64_t)m1.ld * 32u * 4u < (1ull << 31) && (uint64_t)m2.ld * 64u * 4u < (1ull << 31)) {
        const uint64_t wgs = (uint64_t)((M + 127u) / 128u) * ((N + 63u) / 64u) * nmats;
        double best_p = 1e30;
        uint32_t ns_p = 1;
        for (uint32_t ns = 1; ns <= 8 && (ns == 1 || K / ns >= 128); ++ns) {
            if ((double)ns * out_bytes > (double)(512ull << 20)) break;
            const double stages = (double)((K + ns - 1) / ns + 31u) / 32.0;
            const double t = rounds(wgs * ns) * (stages * 1.4 + 2.5) + (ns > 1 ? 4.0 + ns * out_bytes / 7.0e6 : 0.0); // (slab writes are in the per-stage figure)
            if (t < best_p) { best_p = t; ns_p = ns; }
        }
        const int force = ctx->tuning[WG_TUNE_F32_PANELS]; // experiments: 0 = never, 1 = whenever applicable
        // (what the tiled plan's time above leaves out and short-K launches feel: ~3 us per round of workgroups and the output written at ~3.5 TB/s --
        // 1024 x 1024 x 128 x 32 matrices: 59 us by the formula, 107 measured)
        const double tiled = best + rounds(tiles * nmats * nsplit) * 3.0 + (nsplit == 1 && tail_r == 0 ? out_bytes / 3.5e6 : 0.0);
        // (the mid estimate leaves the output's write time out, like `best`: compared with the tiled plan WITHOUT that term)
        const double tiled_m = best + rounds(tiles * nmats * nsplit) * 3.0 + 5.0;
        const bool mid_wins = mid_bm && (mid_knob >= 1 || (force != 1 && mid_est < 0.97 * best_p && mid_est < 0.97 * tiled_m));
        if (!mid_wins && (force >= 0 ? force == 1 : best_p < 0.95 * tiled))
            return wgk_gemm_f32_skinny(ctx, trans, M, N, K, nmats, out, out_ld, out_batch, m1, m2, alpha, beta, 1u, false, ns_p);
    }
    if (mid_bm) {
        const double tiled = best + rounds(tiles * nmats * nsplit) * 3.0 + 5.0; // (+ launch and drain; the output's write time is on neither side)
        if (mid_knob >= 1 || mid_est < 0.97 * tiled)
            return wgk_gemm_f32_mid(ctx, trans, mid_bm, mid_bn, M, N, K, nmats, out, out_ld, out_batch, m1, m2, alpha, beta, ctx->tuning[WG_TUNE_F32_MID_SPLIT] > 1 ? (uint32_t)ctx->tuning[WG_TUNE_F32_MID_SPLIT] : mid_ns);
    }
    float *part = nullptr;
    g.nsplit = nsplit;
    g.k_per_split = nsplit > 1 ? (((K + BK - 1) / BK + nsplit - 1) / nsplit) * BK : K;
    if (nsplit > 1) {
        g.nsplit = nsplit = (K + g.k_per_split - 1) / g.k_per_split; // no empty splits
        void *ws = nullptr;
        if (int rc = wg_ctx_workspace(ctx, (size_t)nsplit * M * N * nmats * sizeof(float), &ws)) return rc;
        part = (float *)ws;
        g.c = part; g.ldc = M; g.c_batch = (uint64_t)M * N; // slab (z, s) at ((z * nsplit + s) * M * N)
        g.alpha = 1.f; g.beta = 0.f;                         // raw partial sums; alpha/beta are applied by the reduce
    }
    if ((uint64_t)nmats * nsplit > 65535) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: nmats * splits exceeds 65535");
    const dim3 grid((uint32_t)tiles, nmats * nsplit), block(kThreads);
    if (tail_r > 0 && nsplit == 1) { // tail split (see the plan above): dense f32 partial tiles in the workspace + an ordered reduce
        const uint32_t r = tail_r, sp = tail_sp;
        const uint32_t kps = ((ktiles + sp - 1) / sp) * BK, n = (K + kps - 1) / kps;
        void *ws = nullptr;
        if (int rc = wg_ctx_workspace(ctx, (size_t)n * r * BM * BN * sizeof(float), &ws)) return rc;
        const uint32_t full = (uint32_t)(tiles * nmats) - r;
        if (nmats > 1) g.flat_tiles = (uint32_t)tiles; // the ids run through the batch
        if (trans) hipLaunchKernelGGL(gemm_f32_kernel<true>, dim3(full, 1), block, 0, ctx->stream, g);
        else hipLaunchKernelGGL(gemm_f32_kernel<false>, dim3(full, 1), block, 0, ctx->stream, g);
        GemmArgs gt = g;
        gt.tile_base = full; gt.tail_tiles = r; gt.nsplit = n; gt.k_per_split = kps;
        gt.c = (float *)ws; gt.c_batch = 0; gt.alpha = 1.f; gt.beta = 0.f;
        if (trans) hipLaunchKernelGGL(gemm_f32_kernel<true>, dim3(r, n), block, 0, ctx->stream, gt);
        else hipLaunchKernelGGL(gemm_f32_kernel<false>, dim3(r, n), block, 0, ctx->stream, gt);
        hipLaunchKernelGGL(gemm_f32_tail_reduce, dim3(r, BN / 4), dim3(256), 0, ctx->stream, gt);
        WG_HIP_TRY(hipGetLastError());
        return WG_OK;
    }
    if (trans) hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, block, 0, ctx->stream, g);
    else hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, block, 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    if (nsplit > 1) return wg_splitk_reduce(ctx, part, nsplit, M, N, nmats, WG_F32, out, out_ld, out_batch, alpha, beta);
    return WG_OK;
}

// out (M x N, column-major) = a (M x K, m-contiguous) * b (K x N, N-CONTIGUOUS: element (k, n) at n + k * ld): the row-major GemmTr in column-major terms (api.hip
// wg_gemm_rm). The 256 x 128 tile kernel with its B_NC tile bodies, whole K per workgroup (no K cut, no tail split: from one round of tiles on that is the plain
// launch's plan anyway). WG_ERR_UNSUPPORTED without a message: not a product this launch takes -- the caller transposes `b` and calls wgk_gemm_f32.
int wgk_gemm_f32_nt(wg_ctx *ctx, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats, float *out, uint32_t out_ld, uint64_t out_batch, wgk_mat a_mcontig, wgk_mat b_ncontig,
                    float alpha, float beta) {
    if (M == 0 || N == 0 || nmats == 0) return WG_OK;
    // (any leading dimension, offset and batch stride: LDS-DMA and the 16-byte loads / stores take element-aligned addresses -- wg_internal.hpp wg_ld_u)
    if (M % 4u || N % 4u || K % 4u || K == 0 || nmats > 65535u) return WG_ERR_UNSUPPORTED;
    GemmArgs g;
    g.a = (const float *)a_mcontig.ptr; g.lda = a_mcontig.ld; g.a_batch = a_mcontig.batch;
    g.b = (const float *)b_ncontig.ptr; g.ldb = b_ncontig.ld; g.b_batch = b_ncontig.batch;
    g.c = out; g.ldc = out_ld; g.c_batch = out_batch;
    g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta;
    g.tile_base = 0; g.tail_tiles = 0; g.out_c = out; g.out_ldc = out_ld; g.out_alpha = alpha; g.out_beta = beta;
    g.flat_tiles = 0; g.out_batch = out_batch;
    g.tiles_m = (M + BM - 1) / BM;
    g.tiles_n = (N + BN - 1) / BN;
    g.nsplit = 1; g.k_per_split = ((K + BK - 1) / BK) * BK;
    // 32-bit DMA offsets within a k-tile: 16 k rows of either operand
    g.dma_ok = ((uint64_t)a_mcontig.ld * 16u * 4u < (1ull << 31)) && ((uint64_t)b_ncontig.ld * 16u * 4u < (1ull << 31)) ? 1u : 0u;
    const uint64_t tiles = (uint64_t)g.tiles_m * g.tiles_n;
    if (tiles > 0x7fffffffull) return WG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((gemm_f32_kernel<false, true>), dim3((uint32_t)tiles, nmats), dim3(kThreads), 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}
