// f16 Gemm, mid-size variant: 128 x 128 block tile, two workgroups per CU.
//
// The 256 x 256 kernel (gemm_f16.hip) owns a whole CU per workgroup; an output with fewer than ~one such tile per CU
// (2048^3: 64 tiles on 256 CUs) can only fill the chip through split-K, whose f32 partial slabs then cost more than the
// product (2048^3: 31 us kernel + 12 us reduce, against 12 us of MFMA work). This kernel trades MFMA efficiency per
// workgroup (64 x 64 per wave: twice the LDS bytes per flop) for four times as many tiles and no partials.
//
//   * 256 threads = 2 x 2 waves of 64 x 64 = 4 x 4 MFMA tiles (v_mfma_f32_16x16x32_f16), f32 accumulation, one RNE rounding;
//   * K advances in half-steps of 32 k. A half-stage (A 128 x 32 + B 32 x 128 = 16 KiB) lives in one of 5 LDS slots = 80 KiB, so two
//     workgroups share a CU (one's barrier / LDS latency is covered by the other's MFMAs) and each keeps up to 4 half-stages
//     (64 KiB) of LDS-DMA in flight: with one stage of lead the kernel ran at the L2 latency (1.3 us per stage), not at its
//     LDS bound. During half-step H a wave multiplies the fragments read during H-1, reads those of H+1 into the other register
//     set, issues its 4 pieces of half-stage H+5 into the slot H-1 released, and ends with lgkmcnt(0) + vmcnt(12) + barrier;
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4) in the swizzled layouts of the big kernel at half-stage granularity
//     (64-byte rows for k-contiguous operands, chunk c at c ^ G(.), conflict-free ds_read_b128; 256-byte [4 k][32 m] blocks +
//     ds_read_b64_tr_b16 + v_permlane16_swap for column-major A), and the same row permutation inside a wave tile so that a
//     lane's accumulators of a tile pair are 8 consecutive rows of C (16-byte stores);
//   * tile order: 4-tile-row strips, contiguous ranges per XCD (ids are dealt round-robin to the 8 XCDs), so an XCD's L2 sees
//     a compact patch of the output.
// M0 is written by the inline asm without save/restore, as in gemm_f16.hip (tests/test_abi_and_host.py checks the ISA).
// Bound: LDS bandwidth (per CU and half-step pair: 64 KiB of fragment reads + 32 KiB of DMA writes against 512 MFMA cycles per
// SIMD) and L2 -> CU bandwidth; see DESIGN.md section 3.
#include "gemm_f16_common.hpp"

namespace wgf16 {
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128, TN = 128, TK = 64;
constexpr int T_A_BYTES = TM * TK * 2;          // 16 KiB
constexpr int T_STAGE = T_A_BYTES + TN * TK * 2; // 32 KiB: [A | B]
constexpr uint32_t T_BIAS = 3072;               // see M16_BIAS in gemm_f16.hip

__device__ __forceinline__ void t_set_m0(uint32_t lds_dst) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_dst)); }
template <int IMM>
__device__ __forceinline__ void t_dma(uint32_t voff, const void *sbase) {
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%c2" ::"v"(voff), "s"(sbase), "i"(IMM));
}

// workgroup id -> tile. Order index o walks strips of 4 tile rows (column by column inside a strip); XCD x = id % 8 takes a
// contiguous range of o.
__device__ __forceinline__ void tile_of128(uint32_t bid, uint32_t tiles_m, uint32_t tiles_n, uint32_t &tm, uint32_t &tn) {
    const uint32_t nwg = tiles_m * tiles_n;
    const uint32_t q = nwg / 8u, r = nwg % 8u, xcd = bid % 8u, local = bid / 8u;
    const uint32_t o = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + local;
    const uint32_t strip = o / (4u * tiles_n);
    const uint32_t within = o - strip * 4u * tiles_n;
    const uint32_t h = min(4u, tiles_m - 4u * strip);
    tn = within / h;
    tm = 4u * strip + (within - tn * h);
}

template <bool TRANS_A>
__global__ __launch_bounds__(256, 2) void gemm_f16_t128_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * T_STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i16 = lane & 15, kg = lane >> 4;
    const int aq = i16 >> 2, bb = i16 & 3;
    const int gq = (4 - aq) & 3; // G(aq), G = {0,3,2,1}

    uint32_t tm, tn;
    tile_of128(blockIdx.x, g.tiles_m, g.tiles_n, tm, tn);
    const uint32_t m0 = tm * TM, n0 = tn * TN;
    const uint32_t z = blockIdx.y / g.nsplit, split = blockIdx.y % g.nsplit;
    const uint32_t k_begin = split * g.k_per_split;
    const uint32_t K_loc = min(g.K - k_begin, g.k_per_split);
    const _Float16 *A = g.a + z * g.a_batch + (TRANS_A ? (uint64_t)k_begin : (uint64_t)k_begin * g.lda);
    const _Float16 *B = g.b + z * g.b_batch + k_begin;

    // ---- DMA addressing (per wave: pieces 4 wave + q of A and of B; a piece = 1 KiB of LDS = 64 lanes x 16 bytes) ----
    // Rows / columns past the end of a ragged tile are clamped to the last valid one (their results are never stored).
    uint32_t a_voff[4], b_voff[4];
    const _Float16 *a_base, *b_base = B + (uint64_t)n0 * g.ldb;
    if constexpr (TRANS_A) a_base = A + (uint64_t)m0 * g.lda; else a_base = A + m0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t P = 4u * wave + q;
        if constexpr (TRANS_A) { // rows 8P..8P+7 of op(A), 128 bytes (64 k) each; chunk swizzle keyed on (row>>3)&3 (rows are read permuted)
            const uint32_t row = 8u * P + (lane >> 3);
            const uint32_t f = ((4u - ((row >> 3) & 3u)) & 3u) | (((row >> 1) & 1u) << 2);
            const uint32_t ra = min(row, g.M - 1u - m0);
            a_voff[q] = (ra * g.lda + 8u * ((lane & 7u) ^ f)) * 2u + (T_BIAS - 1024u * q);
        } else { // half-stage hs = P>>3, k-quad kq = P&7: blocks [kq][mblk = lane>>4] of [4 k][32 m]; k row (lane>>2)&3, 16-byte unit lane&3
            const uint32_t k = 32u * (P >> 3) + 4u * (P & 7u) + ((lane >> 2) & 3u);
            const uint32_t m = min(32u * (lane >> 4) + 8u * (lane & 3u), g.M - 8u - m0); // M % 8 == 0
            a_voff[q] = (k * g.lda + m) * 2u + (T_BIAS - 1024u * q);
        }
        const uint32_t row = 8u * P + (lane >> 3);
        const uint32_t f = ((4u - ((row >> 2) & 3u)) & 3u) | (((row >> 1) & 1u) << 2);
        const uint32_t rb = min(row, g.N - 1u - n0);
        b_voff[q] = (rb * g.ldb + 8u * ((lane & 7u) ^ f)) * 2u + (T_BIAS - 1024u * q);
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(WG_AS3 char *)smem;
    const uint32_t lds_a_wave = __builtin_amdgcn_readfirstlane(lds_base + wave * 4096);
    const uint32_t lds_b_wave = __builtin_amdgcn_readfirstlane(lds_base + T_A_BYTES + wave * 4096);
    auto issue_stage = [&](uint32_t s) { // DMA of stage s into slot s & 1
        const char *ga, *gb = (const char *)(b_base + 64u * s) - T_BIAS;
        if constexpr (TRANS_A) ga = (const char *)(a_base + 64u * s) - T_BIAS;
        else ga = (const char *)(a_base + (uint64_t)(64u * s) * g.lda) - T_BIAS;
        const uint32_t slot = (s & 1u) * T_STAGE;
        t_set_m0(lds_a_wave + slot);
        t_dma<0>(a_voff[0], ga); t_dma<1024>(a_voff[1], ga); t_dma<2048>(a_voff[2], ga); t_dma<3072>(a_voff[3], ga);
        t_set_m0(lds_b_wave + slot);
        t_dma<0>(b_voff[0], gb); t_dma<1024>(b_voff[1], gb); t_dma<2048>(b_voff[2], gb); t_dma<3072>(b_voff[3], gb);
    };

    // ---- per-lane LDS read offsets (within a stage slot) ----
    // B tile u, half-step hs: row 64 wn + 16 u + i16, logical chunk 4 hs + kg
    uint32_t b_off[2];
#pragma unroll
    for (int hs = 0; hs < 2; ++hs)
        b_off[hs] = T_A_BYTES + ((uint32_t)64 * wn + i16) * 128u + (uint32_t)(((kg ^ gq) | ((hs ^ (bb >> 1)) << 2)) * 16);
    // A, TN: MFMA tile t = 2 p + tb, MFMA row i16 <-> tile row 64 wm + 32 p + 8 aq + 4 tb + bb
    uint32_t a_off[2][2]; // TN: [tb][hs];  NN: [0][0] only
    if constexpr (TRANS_A) {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int hs = 0; hs < 2; ++hs)
                a_off[tb][hs] = (64u * wm + 8u * aq + 4u * tb + bb) * 128u + (uint32_t)(((kg ^ gq) | ((hs ^ (bb >> 1)) << 2)) * 16);
    } else {
        // transpose reads: lane row kg reads k-quad 4 (kg>>1) + 2 ins + h, 8-byte half (kg&1) of unit i16 of block (kq, mblk = 2 wm + p)
        a_off[0][0] = (uint32_t)((kg & 2) * 2) * 1024u + (2u * wm) * 256u + (uint32_t)i16 * 16u + (uint32_t)(kg & 1) * 8u;
        a_off[0][1] = a_off[1][0] = a_off[1][1] = 0;
    }

    floatx4 acc[4][4]; // [t][u]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][u][e] = 0.f;

    const uint32_t S = K_loc / 64u; // the launcher guarantees K_loc % 64 == 0, S >= 1
    issue_stage(0);
    for (uint32_t s = 0; s < S; ++s) {
        wait_dma_all();   // stage s has landed (this wave's pieces) ...
        __syncthreads();  // ... everyone's; and every wave is done reading the other slot
        if (s + 1 < S) issue_stage(s + 1);
        const char *sl = smem + (s & 1u) * T_STAGE;
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            uintx4 a_r[4];
            half8_t b_f[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) b_f[u] = lds_h8(sl + b_off[hs] + u * 2048);
            if constexpr (TRANS_A) {
#pragma unroll
                for (int t = 0; t < 4; ++t) a_r[t] = __builtin_bit_cast(uintx4, lds_h8(sl + a_off[t & 1][hs] + (t >> 1) * 4096));
            } else {
                const char *sa = sl + hs * 8192 + a_off[0][0];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { // read i = 2 h + ins lands in tile 2 p + ins, dwords 2 h, 2 h + 1
                        const int h = i >> 1, ins = i & 1;
                        const uintx2 v = __builtin_bit_cast(uintx2, lds_tr(sa + (2 * ins + h) * 1024 + p * 256));
                        a_r[2 * p + ins][2 * h] = v[0];
                        a_r[2 * p + ins][2 * h + 1] = v[1];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) { // put the k-groups back on the lane rows the MFMA expects
                        const uintx2 r = __builtin_amdgcn_permlane16_swap(a_r[2 * p][i], a_r[2 * p + 1][i], false, false);
                        a_r[2 * p][i] = r[0];
                        a_r[2 * p + 1][i] = r[1];
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a_r[t]), b_f[u], acc[t][u], 0, 0, 0);
        }
    }

    // ---- epilogue: lane holds, per (pair p, N tile u), rows 32 p + 8 kg + 0..7 of column 16 u + i16 of the wave tile ----
    const bool full_tile = (m0 + TM <= g.M) && (n0 + TN <= g.N);
    const uint32_t row0 = m0 + 64u * wm + 8u * kg;
    if (g.nsplit > 1) { // split-K: raw f32 partial sums to this split's slab (dense, ld = M); wg_splitk_reduce finishes
        float *P = g.part + ((uint64_t)z * g.nsplit + split) * ((uint64_t)g.M * g.N);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t col = n0 + 64u * wn + 16u * u + i16;
            if (!full_tile && col >= g.N) continue;
            float *pc = P + (uint64_t)col * g.M + row0;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (!(full_tile || row0 + 32 * p < g.M)) continue;
                float4 *d = reinterpret_cast<float4 *>(pc + 32 * p);
                d[0] = make_float4(acc[2 * p][u][0], acc[2 * p][u][1], acc[2 * p][u][2], acc[2 * p][u][3]);
                d[1] = make_float4(acc[2 * p + 1][u][0], acc[2 * p + 1][u][1], acc[2 * p + 1][u][2], acc[2 * p + 1][u][3]);
            }
        }
        return;
    }
    _Float16 *C = g.c + z * g.c_batch;
    const float alpha = g.alpha, beta = g.beta;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t col = n0 + 64u * wn + 16u * u + i16;
        if (!full_tile && col >= g.N) continue;
        _Float16 *cc = C + (uint64_t)col * g.ldc + row0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (!(full_tile || row0 + 32 * p < g.M)) continue; // 8 consecutive rows, all in or all out (M % 8 == 0)
            float r[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                r[q] = acc[2 * p][u][q];
                r[4 + q] = acc[2 * p + 1][u][q];
            }
            if (alpha != 1.f) {
#pragma unroll
                for (int q = 0; q < 8; ++q) r[q] *= alpha;
            }
            if (beta != 0.f) { // beta == 0 never reads C
                const half8_t c = *reinterpret_cast<const half8_t *>(cc + 32 * p);
#pragma unroll
                for (int q = 0; q < 8; ++q) r[q] = fmaf(beta, (float)c[q], r[q]);
            }
            half8_t v;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (_Float16)r[q];
            *reinterpret_cast<half8_t *>(cc + 32 * p) = v;
        }
    }
}

} // namespace

int t128_launch(wg_ctx *ctx, bool trans, dim3 grid, const GemmArgs &g) {
    if (trans) hipLaunchKernelGGL((gemm_f16_t128_kernel<true>), grid, dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL((gemm_f16_t128_kernel<false>), grid, dim3(256), 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

} // namespace wgf16
