// f16 Gemm, previous generation: v_mfma_f32_32x32x16_f16, 5-slot ring of 32-k half-stages (see the comments below), and the
// generic any-shape fallback. Kept in the library: the shipped 16x16x32 kernel (gemm_f16.hip) needs K % 64 == 0 and >= 192 k per
// split; everything else on the MFMA path runs here.
// Bound: MFMA (v_mfma_f32_32x32x16_f16: 2.5 PFLOP/s dense peak). Structure of the fast path:
//   * workgroup = 256 threads = 4 waves, ONE wave per SIMD, each wave owns the whole 512-entry register file:
//     block tile 256(M) x 256(N) x 64(K), wave tile 128 x 128 = 4 x 4 MFMA tiles -> 256 accumulator registers.
//     (LDS reads per K-step: 4 waves x 32 KiB = 128 KiB vs 192 KiB for a 2x4 8-wave split of the same tile.)
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4: 16 B per lane, no VGPR round trip), 2 stages x 64 KiB.
//   * B (K x N, k contiguous) and, for TN, op(A) (k contiguous) are staged as [row][64 k] = 128-byte rows; a lane's
//     MFMA operand (8 consecutive k) is one ds_read_b128. The 16-byte chunk index is XOR-swizzled with (row>>1)&7 so
//     that every 16-lane ds_read_b128 group hits 16 distinct (bank-half, chunk) slots -- conflict-free. The DMA writes
//     LDS lane-linearly, so the swizzle is applied to the per-lane SOURCE address (guide rule 21).
//   * NN: A (M x K) is M-contiguous, but the MFMA wants 8 consecutive k per lane: the tile is staged as 256-byte blocks
//     [k/4][m/32][4 k][32 m] (straight from the global layout: each DMA wave-instruction fetches 4 k-rows x 256 B) and
//     read with ds_read_b64_tr_b16, the gfx950 LDS transpose read: a 16-lane group loads a 4(k) x 16(m) patch and
//     each lane receives 4 consecutive k of ONE m. MFMA row i of tile (T, tb) is mapped to
//         m = 64 T + 32 g + 8 c + 4 (tb ^ g) + e,   i = 16 g + 4 c + e,
//     i.e. two tiles interleave 4-row pieces: (a) the two 16-lane groups of a half-wave read disjoint banks
//     (conflict-free), and (b) in the epilogue a lane's registers of the tile pair are 8 consecutive rows of C = one
//     16-byte store.
//   * workgroup ids are remapped into 16 x 16 super-tiles (the 256 workgroups resident at once), each XCD working
//     on a 4 x 8 patch of it, so that A/B panels are shared in the XCD's L2 and across XCDs in the 256 MiB MALL.
// Ragged M, N (any multiple of 8) stay on this path: DMA source rows are clamped, the epilogue is predicated.
// Shapes it does not cover (M % 8, K % 32, misaligned views) are staged into padded copies by the launcher (gemm_f16.hip) or, when
// tiny, take the small generic kernel below.
#include "gemm_f16_common.hpp"

namespace wgf16 {
namespace {

// NWN = waves along N (2 or 4); always 2 waves along M.  NWN = 2: 4 waves, one per SIMD, 128x128 each (512 registers);
// NWN = 4: 8 waves, two per SIMD, 128x64 each (<= 256 registers).
//
// Pipeline: the K loop advances in HALF-steps of 32 k. A half-stage (A 256x32 + B 32x256 = 32 KiB) lives in one of
// NSLOT = 5 LDS slots (5 x 32 KiB = the whole 160 KiB). During half-step h every wave
//   * runs its MFMAs on slot h%5 (fragments were read one substep earlier -- across the barrier for the first substep),
//   * issues its share of the LDS-DMA of half-stage h+4 into slot (h-1)%5 (released by the barrier that opened h),
//   * and ends with `s_waitcnt vmcnt(2 stages)` + barrier: half-stage h+2 has landed; it was issued 2-3 half-steps
//     (2-3 thousand cycles) earlier, so the wait never sees L2/MALL latency, and no DMA is ever issued "just in time".
template <bool TRANS_A, int NWN>
__global__ __launch_bounds__(128 * NWN, NWN / 2) void gemm_f16_kernel(GemmArgs g) {
    constexpr int NWAVES = 2 * NWN;
    constexpr int NU = 8 / NWN;          // B fragments (32 columns each) per wave: 4 or 2
    constexpr int WN_COLS = 32 * NU;     // columns per wave
    constexpr int PCS = 16 / NWAVES;     // DMA pieces of A (and of B) per wave per half-stage: 4 or 2
    __shared__ __attribute__((aligned(16))) char smem[NSLOT * HSTAGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int hk = lane >> 5;          // which 8 of the MFMA's 16 k this lane feeds
    const int i32 = lane & 31;         // MFMA row (A) / column (B) index
    const int g1 = (lane >> 4) & 1, cq = (lane >> 2) & 3, e4 = lane & 3;

    uint32_t tm, tn;
    tile_of(blockIdx.x, g.tiles_m, g.tiles_n, tm, tn);
    const uint32_t m0 = tm * BM, n0 = tn * BN;
    const uint32_t z = blockIdx.y / g.nsplit, split = blockIdx.y % g.nsplit;
    const uint32_t k_begin = split * g.k_per_split; // multiple of BKH
    const uint32_t K_loc = min(g.K - k_begin, g.k_per_split);
    const _Float16 *A = g.a + z * g.a_batch + (TRANS_A ? (uint64_t)k_begin : (uint64_t)k_begin * g.lda);
    const _Float16 *B = g.b + z * g.b_batch + k_begin;
    _Float16 *C = g.c + z * g.c_batch;

    // ---- DMA addressing: this wave stages pieces P = PCS*wave + q (q < PCS) of A and of B; 1 KiB per piece ----
    // B half-tile (and op(A) for TN), [row][32 k] = 64-byte rows: piece P = rows 16P..16P+15, lane -> row 16P + (lane>>2),
    //   position lane&3 holds logical 16-byte chunk (lane&3) ^ ((row>>2)&3) = (lane&3) ^ (lane>>4)
    // A half-tile (NN), 256-byte blocks [kq][mblk][4 k][32 m]: piece P = blocks 4P..4P+3 -> kq = P>>1, mblk = 4*(P&1) + (lane>>4),
    //   k row within the block (lane>>2)&3, 16-byte piece lane&3
    uint32_t a_voff[PCS], b_voff[PCS]; // 32-bit per-lane byte offsets relative to wave-uniform base pointers
    const _Float16 *a_base, *b_base = B + (uint64_t)n0 * g.ldb;
    if constexpr (TRANS_A) a_base = A + (uint64_t)m0 * g.lda; else a_base = A + m0;
#pragma unroll
    for (int q = 0; q < PCS; ++q) {
        const uint32_t P = PCS * wave + q;
        const int chunk = (lane & 3) ^ (lane >> 4);
        const uint32_t row = 16u * P + (lane >> 2);
        // ragged edge tiles: rows / row-pieces past the end of the matrix are CLAMPED to the last valid one (valid memory,
        // results discarded by the predicated epilogue) -- computed once here, so the K loop is identical for every tile
        const uint32_t rb = min(row, g.N - 1u - n0);
        b_voff[q] = (rb * g.ldb + 8u * chunk) * 2u;
        if constexpr (TRANS_A) {
            const uint32_t ra = min(row, g.M - 1u - m0);
            a_voff[q] = (ra * g.lda + 8u * chunk) * 2u;
        } else {
            const uint32_t mpiece = min(128u * (P & 1) + 32u * (lane >> 4) + 8u * (lane & 3), g.M - 8u - m0); // M % 8 == 0
            a_voff[q] = ((4u * (P >> 1) + ((lane >> 2) & 3)) * g.lda + mpiece) * 2u;
        }
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(WG_AS3 char *)smem;
    // point p (0 .. 2*PCS-1) of a half-step issues ONE DMA piece: A piece p>>1 when p is even, B piece p>>1 when odd
    auto stage_piece = [&](uint32_t slot, uint32_t k0, int p) {
        const uint32_t sa = __builtin_amdgcn_readfirstlane(lds_base + slot * HSTAGE_BYTES + wave * (PCS * 1024));
        const int q = p >> 1;
        if ((p & 1) == 0) {
            if constexpr (TRANS_A) glds16s<0>(a_voff[q], a_base + k0, sa + q * 1024);
            else glds16s<0>(a_voff[q], a_base + (uint64_t)k0 * g.lda, sa + q * 1024);
        } else {
            glds16s<0>(b_voff[q], b_base + k0, sa + HA_BYTES + q * 1024);
        }
    };

    // ---- per-lane LDS read offsets (within a slot) ----
    // B fragment for N-tile u, substep kk (0/1): row n = WN_COLS*wn + 32 u + i32, chunk (2kk + hk) ^ ((i32>>2)&3)
    uint32_t b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
        b_off[kk] = HA_BYTES + ((uint32_t)WN_COLS * wn + i32) * 64u + (((2 * kk + hk) ^ ((i32 >> 2) & 3)) * 16u);
    // A fragment.  MFMA row i32 = 16 g1 + 4 cq + e4 of tile (T, tb)  <->  m = 128 wm + 64 T + 32 g1 + 8 cq + 4 (tb ^ g1) + e4
    uint32_t a_off[2][2]; // TN: [tb][kk] ; NN: [tb][0] only
    if constexpr (TRANS_A) {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const uint32_t ml = 128u * wm + 32u * g1 + 8u * cq + 4u * (tb ^ g1) + e4;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) a_off[tb][kk] = ml * 64u + (((2 * kk + hk) ^ ((ml >> 2) & 3)) * 16u);
        }
    } else {
        // transpose read: source lane p = lane&15 supplies 4 consecutive m at k row (p>>2): the address is linear in p
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
            a_off[tb][0] = hk * 4096u + (4u * wm + g1) * 256u + (uint32_t)(lane & 15) * 16u + (tb ^ g1) * 8u;
    }

    floatx16 acc[2][2][NU]; // [T][tb][u]
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[T][tb][u][e] = 0.f;

    // ---- hand-placed schedule ---------------------------------------------------------------------------------------
    // One "slot" per MFMA; __builtin_amdgcn_sched_barrier(0) between slots pins the ORDER, the compiler still inserts
    // the (counted) lgkmcnt waits, so ds_read -> MFMA dependencies stay correct by construction.
    //   slot (kk, j): MFMA #j of substep kk (A fragment j / NU, B fragment j % NU)
    //                 + its share of the LDS reads of the NEXT substep's fragments (other register set; for kk = 1 that is
    //                   substep 0 of the next half-stage, already published by the barrier that opened this half-step)
    //                 + every 4th slot: one DMA piece of half-stage h+4
    short4_t a_lo[2][4], a_hi[2][4]; // NN: the two transpose reads of each A fragment
    half8_t a_f[2][4];               // TN: A fragment by ds_read_b128
    half8_t b_f[2][NU];
    constexpr int kReads = (TRANS_A ? 4 : 8) + NU;
    constexpr int kSlots = 4 * NU; // MFMAs per substep
    static_assert(2 * kSlots == 8 * PCS, "one DMA point per 4 MFMA slots");

    auto read_op = [&](const char *s, int kk, int r, int set) {
        if constexpr (TRANS_A) {
            if (r == 0) a_f[set][0] = lds_h8(s + a_off[0][kk]);
            else if (r <= NU) b_f[set][r - 1] = lds_h8(s + b_off[kk] + (r - 1) * 2048);
            else { const int f = r - NU; a_f[set][f] = lds_h8(s + a_off[f & 1][kk] + (f >> 1) * 4096); } // f = (T = f>>1, tb = f&1)
        } else {
            if (r < 2) { // A fragment 0 first, then all of B, then the rest of A: the first MFMA's operands arrive first
                const char *p = s + a_off[0][0] + kk * 8192;
                if (r == 0) a_lo[set][0] = lds_tr(p); else a_hi[set][0] = lds_tr(p + 2048);
            } else if (r < 2 + NU) {
                b_f[set][r - 2] = lds_h8(s + b_off[kk] + (r - 2) * 2048);
            } else {
                const int f = 1 + ((r - 2 - NU) >> 1); // fragment f = (T = f>>1, tb = f&1)
                const char *p = s + a_off[f & 1][0] + kk * 8192 + (f >> 1) * 512;
                if (((r - 2 - NU) & 1) == 0) a_lo[set][f] = lds_tr(p); else a_hi[set][f] = lds_tr(p + 2048);
            }
        }
    };

    // one half-step on LDS slot `cur`; DMA (if any) goes to slot `dst`, next half-stage's first fragments come from `nxt`
    auto half_step = [&](uint32_t cur, uint32_t nxt, uint32_t dst, uint32_t k_dma, auto do_dma, auto has_next) {
        const char *s = smem + cur * HSTAGE_BYTES;
        const char *sn = smem + nxt * HSTAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int j = 0; j < kSlots; ++j) {
                const int f = j / NU, u = j % NU;
                half8_t af;
                if constexpr (TRANS_A) af = a_f[kk][f]; else af = cat(a_lo[kk][f], a_hi[kk][f]);
                acc[f >> 1][f & 1][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b_f[kk][u], acc[f >> 1][f & 1][u], 0, 0, 0);
                if (kk == 0) {
#pragma unroll
                    for (int r = j; r < kReads; r += kSlots) read_op(s, 1, r, 1);
                } else if constexpr (decltype(has_next)::value) {
#pragma unroll
                    for (int r = j; r < kReads; r += kSlots) read_op(sn, 0, r, 0);
                }
                if constexpr (decltype(do_dma)::value) {
                    if ((j & 3) == 3) stage_piece(dst, k_dma, (kk * kSlots + j) >> 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto next_slot = [](uint32_t s) { return s + 1 == NSLOT ? 0u : s + 1; };

    const uint32_t nh = K_loc / BKH; // half-steps
    // prologue: up to 4 half-stages in flight, everything landed before the first barrier (once per tile)
    for (uint32_t h = 0; h < 4 && h < nh; ++h) {
#pragma unroll
        for (int p = 0; p < 2 * PCS; ++p) stage_piece(h, h * BKH, p);
    }
    wait_dma_all();
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kReads; ++r) read_op(smem, 0, r, 0);
    __builtin_amdgcn_sched_barrier(0);

    uint32_t cur = 0, h = 0;
    // steady state: half-stage h+4 exists
    for (; h + 4 < nh; ++h) {
        const uint32_t nxt = next_slot(cur);
        const uint32_t dst = cur == 0 ? NSLOT - 1 : cur - 1; // slot of half-stage h-1 == slot of h+4
        half_step(cur, nxt, dst, (h + 4) * BKH, std::true_type{}, std::true_type{});
        wait_dma_keep<2 * 2 * PCS>(); // half-stages h+3 and h+4 may still be in flight; h+2 has landed
        if (!(WG_ABLATE & 1)) __builtin_amdgcn_s_barrier();
        cur = nxt;
    }
    // tail: nothing left to issue
    for (; h + 1 < nh; ++h) {
        const uint32_t nxt = next_slot(cur);
        half_step(cur, nxt, 0, 0, std::false_type{}, std::true_type{});
        wait_dma_all();
        if (!(WG_ABLATE & 1)) __builtin_amdgcn_s_barrier();
        cur = nxt;
    }
    half_step(cur, cur, 0, 0, std::false_type{}, std::false_type{});

    // ---- epilogue: f32 -> f16 (RNE), 16-byte stores. C/D map of the 32x32 MFMA: register e -> row (e&3) + 8 (e>>2) + 4 hk ----
    // rows of tile (T, tb): m = 64 T + 16 gq + 8 hk + 4 (tb ^ (gq>>1)) + (e&3)  => the pair (tb = gq>>1, tb = 1 - (gq>>1)) is 8 consecutive rows
    const bool full_tile = (m0 + BM <= g.M) && (n0 + BN <= g.N); // workgroup-uniform
    if (g.nsplit > 1) { // split-K: raw f32 partial sums to this split's slab (dense, ld = M); wg_splitk_reduce finishes the job
        float *P = g.part + ((uint64_t)z * g.nsplit + split) * ((uint64_t)g.M * g.N);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const uint32_t col = n0 + (uint32_t)WN_COLS * wn + 32u * u + i32;
            if (!full_tile && col >= g.N) continue;
            const uint32_t row0 = m0 + 128u * wm + 8u * hk;
            float *pc = P + (uint64_t)col * g.M + row0;
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int lo = gq >> 1, hi = 1 - lo;
                    if (full_tile || row0 + 64 * T + 16 * gq < g.M) {
                        float4 *d = reinterpret_cast<float4 *>(pc + 64 * T + 16 * gq);
                        d[0] = make_float4(acc[T][lo][u][4 * gq], acc[T][lo][u][4 * gq + 1], acc[T][lo][u][4 * gq + 2], acc[T][lo][u][4 * gq + 3]);
                        d[1] = make_float4(acc[T][hi][u][4 * gq], acc[T][hi][u][4 * gq + 1], acc[T][hi][u][4 * gq + 2], acc[T][hi][u][4 * gq + 3]);
                    }
                }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const uint32_t col = n0 + (uint32_t)WN_COLS * wn + 32u * u + i32;
        if (!full_tile && col >= g.N) continue;
        const uint32_t row0 = m0 + 128u * wm + 8u * hk;
        _Float16 *cc = C + (uint64_t)col * g.ldc + row0;
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int lo = gq >> 1, hi = 1 - lo;
                if (!(full_tile || row0 + 64 * T + 16 * gq < g.M)) continue; // 8 consecutive rows, all in or all out (M % 8 == 0)
                float r[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    r[q] = acc[T][lo][u][4 * gq + q];
                    r[4 + q] = acc[T][hi][u][4 * gq + q];
                }
                if (g.alpha != 1.f) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) r[q] *= g.alpha;
                }
                if (g.beta != 0.f) { // beta == 0 never reads C
                    const half8_t c = *reinterpret_cast<const half8_t *>(cc + 64 * T + 16 * gq);
#pragma unroll
                    for (int q = 0; q < 8; ++q) r[q] = fmaf(g.beta, (float)c[q], r[q]);
                }
                half8_t v;
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (_Float16)r[q];
                *reinterpret_cast<half8_t *>(cc + 64 * T + 16 * gq) = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// generic path: any M, N, K % 4 == 0 (the vec4 precondition), any stride/offset the API admits. 64x64 tile, f32 FMA.
// Same numerics contract (exact f16 products, f32 accumulation, one rounding); only the summation order differs.
// ---------------------------------------------------------------------------------------------------------------
template <bool TRANS_A>
__global__ __launch_bounds__(256) void gemm_f16_generic_kernel(GemmArgs g) {
    __shared__ float As[16][65];
    __shared__ float Bs[16][65];
    const uint32_t z = blockIdx.z;
    const _Float16 *A = g.a + z * g.a_batch;
    const _Float16 *B = g.b + z * g.b_batch;
    _Float16 *C = g.c + z * g.c_batch;
    const uint32_t m0 = blockIdx.x * 64u, n0 = blockIdx.y * 64u;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4; // thread owns rows 4tx..4tx+3, cols 4ty..4ty+3
    float acc[4][4] = {};
    for (uint32_t k0 = 0; k0 < g.K; k0 += 16u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = threadIdx.x + 256 * r; // 1024 elements per tile
            {
                const uint32_t kk = TRANS_A ? (f & 15) : (f >> 6), mm = TRANS_A ? (f >> 4) : (f & 63);
                const uint32_t m = m0 + mm, k = k0 + kk;
                float v = 0.f;
                if (m < g.M && k < g.K) v = (float)(TRANS_A ? A[(uint64_t)m * g.lda + k] : A[(uint64_t)k * g.lda + m]);
                As[kk][mm] = v;
            }
            {
                const uint32_t kk = f & 15, nn = f >> 4;
                const uint32_t n = n0 + nn, k = k0 + kk;
                Bs[kk][nn] = (n < g.N && k < g.K) ? (float)B[(uint64_t)n * g.ldb + k] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[q] = As[kk][4 * tx + q]; b[q] = Bs[kk][4 * ty + q]; }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[p][q] = fmaf(a[p], b[q], acc[p][q]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t n = n0 + 4 * ty + q;
        if (n >= g.N) continue;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t m = m0 + 4 * tx + p;
            if (m < g.M) {
                float r = g.alpha == 1.f ? acc[p][q] : g.alpha * acc[p][q];
                if (g.beta != 0.f) r = fmaf(g.beta, (float)C[(uint64_t)n * g.ldc + m], r);
                C[(uint64_t)n * g.ldc + m] = (_Float16)r;
            }
        }
    }
}


} // namespace

#ifndef WG_F16_NWN
#define WG_F16_NWN 2 // 2 = 4 waves, one per SIMD (measured 1198 TF at 8192^3, less LDS traffic); 4 = 8 waves, two per SIMD (1184 TF)
#endif
int legacy_launch(wg_ctx *ctx, bool trans, dim3 grid, const GemmArgs &g) {
    const dim3 block(128 * WG_F16_NWN);
    if (trans) hipLaunchKernelGGL((gemm_f16_kernel<true, WG_F16_NWN>), grid, block, 0, ctx->stream, g);
    else hipLaunchKernelGGL((gemm_f16_kernel<false, WG_F16_NWN>), grid, block, 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}
int generic_launch(wg_ctx *ctx, bool trans, dim3 grid, const GemmArgs &g) {
    if (trans) hipLaunchKernelGGL(gemm_f16_generic_kernel<true>, grid, dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL(gemm_f16_generic_kernel<false>, grid, dim3(256), 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

} // namespace wgf16
