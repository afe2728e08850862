// f16 Gemm (extension -- the reference has no f16 kernel): out = m1 * m2 (NN) or m1^T * m2 (TN), column-major,
// f16 operands, f32 accumulation on the matrix cores, ONE rounding (RNE) to f16 at the end.
//
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
// Shapes the fast path does not cover (M, N % 256, K % 64, misaligned views) take a small generic kernel.
#include "wg_internal.hpp"

#include <type_traits>

namespace {

#ifndef WG_ABLATE
#define WG_ABLATE 0 // timing experiments only: 1 = no barrier, 2 = no DMA, 4 = no LDS reads (bitmask); results are garbage
#endif

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define WG_AS1 __attribute__((address_space(1)))
#define WG_AS3 __attribute__((address_space(3)))

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kThreads = 256;
constexpr int A_BYTES = BM * BK * 2; // 32 KiB
constexpr int B_BYTES = BN * BK * 2;
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;

struct GemmArgs {
    const _Float16 *a; uint32_t lda; uint64_t a_batch;
    const _Float16 *b; uint32_t ldb; uint64_t b_batch;
    _Float16 *c; uint32_t ldc; uint64_t c_batch;
    uint32_t M, N, K;
    uint32_t tiles_m, tiles_n;
};

// LDS-DMA: 16 bytes per lane from `gsrc` (per-lane) to LDS byte address `lds_dst` + 16*lane (`lds_dst` wave-uniform).
// Issued through inline asm ON PURPOSE: hipcc cannot tell that the DMA into stage t+1 does not alias the ds_reads of
// stage t (one LDS array, no alias scopes) and would put `s_waitcnt vmcnt(0)` in front of the first ds_read of every
// K-step, serialising HBM latency with the MFMAs. Hidden from its scoreboard, the DMA stays in flight during the whole
// K-step; the kernel waits for it itself (wait_dma) right before the barrier that publishes the stage.
__device__ __forceinline__ void glds16(const _Float16 *gsrc, uint32_t lds_dst) {
    if (WG_ABLATE & 2) return;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst));
}
// Same DMA with the cheaper address form: wave-uniform 64-bit base in SGPRs + one 32-bit per-lane byte offset + immediate.
template <int IMM>
__device__ __forceinline__ void glds16s(uint32_t voff, const void *sbase, uint32_t lds_dst) {
    if (WG_ABLATE & 2) return;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%c4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst), "i"(IMM));
}
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ short4_t lds_tr(const char *p) {
    if (WG_ABLATE & 4) { short4_t v = { (short)(uintptr_t)p, 1, 2, 3 }; return v; }
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((WG_AS3 short4_t *)p);
}
__device__ __forceinline__ half8_t lds_h8(const char *p) {
    if (WG_ABLATE & 4) { half8_t v = { (_Float16)(float)(uintptr_t)p, 1, 2, 3, 4, 5, 6, 7 }; return v; }
    return *reinterpret_cast<const half8_t *>(p);
}
__device__ __forceinline__ half8_t cat(short4_t lo, short4_t hi) {
    short8_t v = { lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3] };
    return __builtin_bit_cast(half8_t, v);
}

// workgroup id -> (tile_m, tile_n)
__device__ __forceinline__ void tile_of(uint32_t bid, uint32_t tiles_m, uint32_t tiles_n, uint32_t &tm, uint32_t &tn) {
    if ((tiles_m % 16u) == 0 && (tiles_n % 16u) == 0) {
        // 256 consecutive ids = one 16x16 super-tile; hardware deals ids round-robin to the 8 XCDs: XCD x gets a 4x8 patch
        const uint32_t super = bid >> 8, within = bid & 255u;
        const uint32_t xcd = within & 7u, local = within >> 3;
        const uint32_t sm = super % (tiles_m / 16u), sn = super / (tiles_m / 16u);
        tm = sm * 16u + (xcd & 3u) * 4u + (local & 3u);
        tn = sn * 16u + (xcd >> 2) * 8u + (local >> 2);
    } else {
        const uint32_t nwg = tiles_m * tiles_n;
        const uint32_t q = nwg / 8u, r = nwg % 8u, xcd = bid % 8u, local = bid / 8u;
        const uint32_t id = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + local;
        tm = id % tiles_m;
        tn = id / tiles_m;
    }
}

template <bool TRANS_A>
__global__ __launch_bounds__(kThreads, 1) void gemm_f16_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int hk = lane >> 5;          // which 8 of the MFMA's 16 k this lane feeds
    const int i32 = lane & 31;         // MFMA row (A) / column (B) index
    const int g1 = (lane >> 4) & 1, cq = (lane >> 2) & 3, e4 = lane & 3;

    uint32_t tm, tn;
    tile_of(blockIdx.x, g.tiles_m, g.tiles_n, tm, tn);
    const uint32_t m0 = tm * BM, n0 = tn * BN;
    const uint32_t z = blockIdx.y;
    const _Float16 *A = g.a + z * g.a_batch;
    const _Float16 *B = g.b + z * g.b_batch;
    _Float16 *C = g.c + z * g.c_batch;

    // ---- per-lane global source pointers for the DMA pieces (this wave stages pieces P = 8*wave + q, q < 8) ----
    // B tile, [n][64 k] rows: piece P = rows 8P..8P+7, lane -> row 8P + (lane>>3), swizzled chunk lane&7
    const _Float16 *b_src[2]; // q even / odd share everything but the row parity class used by the swizzle
    {
        // row = 8P + (lane>>3) = 64*wave + 8q + (lane>>3); (row>>1)&7 = (4q + (lane>>4)) & 7 = 4*(q&1) + (lane>>4)
        for (int par = 0; par < 2; ++par) {
            const int sw = 4 * par + (lane >> 4);
            const int chunk = (lane & 7) ^ sw;
            const uint32_t row = 64u * wave + 8u * par + (lane >> 3);
            b_src[par] = B + (uint64_t)(n0 + row) * g.ldb + 8u * chunk;
        }
    }
    const _Float16 *a_src[2];
    if constexpr (TRANS_A) {
        for (int par = 0; par < 2; ++par) {
            const int sw = 4 * par + (lane >> 4);
            const int chunk = (lane & 7) ^ sw;
            const uint32_t row = 64u * wave + 8u * par + (lane >> 3);
            a_src[par] = A + (uint64_t)(m0 + row) * g.lda + 8u * chunk;
        }
    } else {
        // A tile, blocks [kq][mblk][4 k][32 m]: piece P = blocks 4P..4P+3 -> kq = P>>1, mblk = 4*(P&1) + (lane>>4),
        // k row within block (lane>>2)&3, 16-byte piece lane&3.  k = 16*wave + 4*(q>>1) + j, m = 128*(q&1) + 32*(lane>>4) + 8*(lane&3)
        for (int par = 0; par < 2; ++par)
            a_src[par] = A + (uint64_t)(16u * wave + ((lane >> 2) & 3)) * g.lda + m0 + 128u * par + 32u * (lane >> 4) + 8u * (lane & 3);
    }

    // 32-bit per-lane byte offsets relative to the tile's (wave-uniform) base pointers
    uint32_t a_voff[8], b_voff[8]; // NN uses a_voff[0..3] (+256-byte immediate for odd pieces); TN/B: [par*4 + (q>>1)]
    const _Float16 *a_base, *b_base = B + (uint64_t)n0 * g.ldb;
    if constexpr (TRANS_A) a_base = A + (uint64_t)m0 * g.lda; else a_base = A + m0;
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int qh = 0; qh < 4; ++qh) {
            const int chunk = (lane & 7) ^ (4 * par + (lane >> 4));
            const uint32_t row = 64u * wave + 8u * par + 16u * qh + (lane >> 3);
            b_voff[par * 4 + qh] = (row * g.ldb + 8u * chunk) * 2u;
            if constexpr (TRANS_A) a_voff[par * 4 + qh] = (row * g.lda + 8u * chunk) * 2u;
        }
    if constexpr (!TRANS_A) {
#pragma unroll
        for (int qh = 0; qh < 4; ++qh)
            a_voff[qh] = ((16u * wave + 4u * qh + ((lane >> 2) & 3)) * g.lda + 32u * (lane >> 4) + 8u * (lane & 3)) * 2u;
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(WG_AS3 char *)smem;
    // quarter `part` (0..3) of this wave's 16 DMA pieces of one stage: 2 of A + 2 of B
    auto stage_part = [&](int buf, uint32_t k0, int part) {
        const uint32_t sa = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_BYTES + wave * 8192);
        const uint32_t sb = sa + A_BYTES;
#pragma unroll
        for (int q = 2 * part; q < 2 * part + 2; ++q) {
            if constexpr (TRANS_A) glds16(a_src[q & 1] + (uint64_t)(16u * (q >> 1)) * g.lda + k0, sa + q * 1024);
            else glds16(a_src[q & 1] + (uint64_t)(k0 + 4u * (q >> 1)) * g.lda, sa + q * 1024);
            glds16(b_src[q & 1] + (uint64_t)(16u * (q >> 1)) * g.ldb + k0, sb + q * 1024);
        }
    };
    auto stage = [&](int buf, uint32_t k0) {
#pragma unroll
        for (int part = 0; part < 4; ++part) stage_part(buf, k0, part);
    };
    // point p (0..15) of a K-step issues ONE DMA piece: A piece p>>1 when p is even, B piece p>>1 when odd
    auto stage_piece = [&](int buf, uint32_t k0, int p, int) {
        const uint32_t sa = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_BYTES + wave * 8192);
        const int q = p >> 1;
        if ((p & 1) == 0) {
            if constexpr (TRANS_A) glds16s<0>(a_voff[(q & 1) * 4 + (q >> 1)], a_base + k0, sa + q * 1024);
            // the instruction's immediate offset is added to the LDS address as well as to the global one: take it back out of M0
            else if (q & 1) glds16s<256>(a_voff[q >> 1], a_base + (uint64_t)k0 * g.lda, sa + q * 1024 - 256);
            else glds16s<0>(a_voff[q >> 1], a_base + (uint64_t)k0 * g.lda, sa + q * 1024);
        } else {
            glds16s<0>(b_voff[(q & 1) * 4 + (q >> 1)], b_base + k0, sa + A_BYTES + q * 1024);
        }
    };

    // ---- per-lane LDS read offsets ----
    // B fragment for N-tile u, k-substep kk: row n = 128 wn + 32 u + i32, chunk (2kk + hk) ^ ((i32>>1)&7)
    uint32_t b_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) b_off[kk] = A_BYTES + (128u * wn + i32) * 128u + (((2 * kk + hk) ^ ((i32 >> 1) & 7)) * 16u);
    // A fragment.  MFMA row i32 = 16 g1 + 4 cq + e4 of tile (T, tb)  <->  m = 128 wm + 64 T + 32 g1 + 8 cq + 4 (tb ^ g1) + e4
    uint32_t a_off[2][4]; // TN: [tb][kk] ; NN: [tb][0] only
    if constexpr (TRANS_A) {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const uint32_t ml = 128u * wm + 32u * g1 + 8u * cq + 4u * (tb ^ g1) + e4;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) a_off[tb][kk] = ml * 128u + (((2 * kk + hk) ^ ((ml >> 1) & 7)) * 16u);
        }
    } else {
        // transpose read: source lane p = lane&15 supplies 4 consecutive m at k row (p>>2): address is linear in p
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
            a_off[tb][0] = hk * 4096u + (4u * wm + g1) * 256u + (uint32_t)(lane & 15) * 16u + (tb ^ g1) * 8u;
    }

    floatx16 acc[2][2][4]; // [T][tb][u]
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[T][tb][u][e] = 0.f;

    // ---- hand-placed K-step schedule ----------------------------------------------------------------------------
    // 64 "slots" per K-step, one per MFMA; __builtin_amdgcn_sched_barrier(0) between slots pins the ORDER, the compiler
    // still inserts the (counted) lgkmcnt waits, so ds_read -> MFMA dependencies stay correct by construction.
    //   slot (kk, j): MFMA #j of substep kk  (A fragment j>>2, B fragment j&3)
    //                 + LDS read #j (j < 12) of substep kk+1's fragments into the other register set
    //                 + (j & 3) == 3: one A and one B DMA piece of the NEXT stage
    // so every LDS read / DMA issue sits in the shadow of a 32-cycle MFMA, a whole substep (16 MFMAs) ahead of its use.
    short4_t a_lo[2][4], a_hi[2][4]; // NN: the two transpose reads of each A fragment
    half8_t a_f[2][4];               // TN: A fragment by ds_read_b128
    half8_t b_f[2][4];

    auto read_op = [&](const char *s, int kk, int r, int set) { // r: 0..11 (NN) / 0..7 (TN)
        if constexpr (TRANS_A) {
            if (r < 4) a_f[set][r] = lds_h8(s + a_off[r & 1][kk] + (r >> 1) * 8192); // fragment r = (T = r>>1, tb = r&1)
            else b_f[set][r - 4] = lds_h8(s + b_off[kk] + (r - 4) * 4096);
        } else {
            if (r < 2) { // A fragment 0 first, then all of B, then the rest of A: the first MFMA's operands arrive first
                const char *p = s + a_off[0][0] + kk * 8192;
                if (r == 0) a_lo[set][0] = lds_tr(p); else a_hi[set][0] = lds_tr(p + 2048);
            } else if (r < 6) {
                b_f[set][r - 2] = lds_h8(s + b_off[kk] + (r - 2) * 4096);
            } else {
                const int f = 1 + ((r - 6) >> 1); // fragment f = (T = f>>1, tb = f&1)
                const char *p = s + a_off[f & 1][0] + kk * 8192 + (f >> 1) * 512;
                if (((r - 6) & 1) == 0) a_lo[set][f] = lds_tr(p); else a_hi[set][f] = lds_tr(p + 2048);
            }
        }
    };
    constexpr int kReads = TRANS_A ? 8 : 12;

    auto compute = [&](int buf, auto prefetch, uint32_t k_next) {
        const char *s = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int r = 0; r < kReads; ++r) read_op(s, 0, r, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int f = j >> 2, u = j & 3;
                half8_t af;
                if constexpr (TRANS_A) af = a_f[cur][f]; else af = cat(a_lo[cur][f], a_hi[cur][f]);
                acc[f >> 1][f & 1][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b_f[cur][u], acc[f >> 1][f & 1][u], 0, 0, 0);
                if (kk < 3 && j < kReads) read_op(s, kk + 1, j, nxt);
                if constexpr (decltype(prefetch)::value) {
                    if ((j & 3) == 3) stage_piece(buf ^ 1, k_next, kk * 4 + (j >> 2) - 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    const uint32_t nk = g.K / BK;
    stage(0, 0);
    for (uint32_t t = 0; t + 1 < nk; ++t) {
        wait_dma();      // this wave's pieces of tile t have landed (issued during the previous K-step)
        if (!(WG_ABLATE & 1)) __syncthreads(); // ... and everybody's; and everyone has left buffer (t+1)&1
        compute(t & 1, std::true_type{}, (t + 1) * BK);
    }
    wait_dma();
    __syncthreads();
    compute((nk - 1) & 1, std::false_type{}, 0);

    // ---- epilogue: f32 -> f16 (RNE), 16-byte stores. C/D map of the 32x32 MFMA: register e -> row (e&3) + 8 (e>>2) + 4 hk ----
    // rows of tile (T, tb): m = 64 T + 16 gq + 8 hk + 4 (tb ^ (gq>>1)) + (e&3)  => the pair (tb = gq>>1, tb = 1 - (gq>>1)) is 8 consecutive rows
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        _Float16 *cc = C + (uint64_t)(n0 + 128u * wn + 32u * u + i32) * g.ldc + m0 + 128u * wm + 8u * hk;
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int lo = gq >> 1, hi = 1 - lo;
                half8_t v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = (_Float16)acc[T][lo][u][4 * gq + q];
                    v[4 + q] = (_Float16)acc[T][hi][u][4 * gq + q];
                }
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
            if (m < g.M) C[(uint64_t)n * g.ldc + m] = (_Float16)acc[p][q];
        }
    }
}

} // namespace

int wgk_gemm_f16(wg_ctx *ctx, bool trans, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats,
                 __half *out, uint32_t out_ld, uint64_t out_batch, wgk_mat m1, wgk_mat m2) {
    if (M == 0 || N == 0 || nmats == 0) return WG_OK;
    if (nmats > 65535) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: more than 65535 matrices in one call");
    GemmArgs g;
    g.a = (const _Float16 *)m1.ptr; g.lda = m1.ld; g.a_batch = m1.batch;
    g.b = (const _Float16 *)m2.ptr; g.ldb = m2.ld; g.b_batch = m2.batch;
    g.c = (_Float16 *)out; g.ldc = out_ld; g.c_batch = out_batch;
    g.M = M; g.N = N; g.K = K;

    auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
    const bool batch_ok = nmats == 1 || (m1.batch % 8 == 0 && m2.batch % 8 == 0 && out_batch % 8 == 0);
    const bool fast = (M % BM == 0) && (N % BN == 0) && (K % BK == 0) && K >= (uint32_t)BK && (m1.ld % 8 == 0) && (m2.ld % 8 == 0) &&
                      (out_ld % 8 == 0) && al16(m1.ptr) && al16(m2.ptr) && al16(out) && batch_ok;
    if (fast) {
        g.tiles_m = M / BM;
        g.tiles_n = N / BN;
        const uint64_t tiles = (uint64_t)g.tiles_m * g.tiles_n;
        if (tiles > 0x7fffffffull) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: too many tiles");
        const dim3 grid((uint32_t)tiles, nmats), block(kThreads);
        if (trans) hipLaunchKernelGGL(gemm_f16_kernel<true>, grid, block, 0, ctx->stream, g);
        else hipLaunchKernelGGL(gemm_f16_kernel<false>, grid, block, 0, ctx->stream, g);
    } else {
        g.tiles_m = (M + 63) / 64;
        g.tiles_n = (N + 63) / 64;
        if (g.tiles_n > 65535) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: N too large for the generic f16 path");
        const dim3 grid(g.tiles_m, g.tiles_n, nmats), block(256);
        if (trans) hipLaunchKernelGGL(gemm_f16_generic_kernel<true>, grid, block, 0, ctx->stream, g);
        else hipLaunchKernelGGL(gemm_f16_generic_kernel<false>, grid, block, 0, ctx->stream, g);
    }
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}
