// f32 Gemm, mid-size outputs: out = m1 * m2 (NN) or m1^T * m2 (TN), column-major, batched   (wgebra gemm.wgsl:28-200, the same contract
// as gemm_f32.hip)
//
// Why a second tile family. gemm_f32.hip's 256 x 128 tile gives 2048^3 only 128 workgroups for 256 CUs: it fills the chip there by cutting K
// (f32 partial slabs + an ordered reduce), and 1024^3 (32 tiles) by a detour over the few-column kernel's 64-column panels -- 13-38 % behind the
// vendor library on 1024^3 .. 2048^3 and on batches of mid-size matrices (profiles/r03_gemm_sweep_full.txt). The f32 matrix cores are slow
// (64 FLOP/clk/SIMD: a 32x32x2 MFMA holds its SIMD for 64 cycles), so a SMALL tile loses nothing per flop -- LDS and L2 bandwidth per flop are a
// sixteenth of what the f16 kernels need -- and simply gives more tiles: 128 x 128 (256 tiles at 2048^3: one per CU), 128 x 64 (512: two
// co-resident per CU), 64 x 64 (256 at 1024^3). No K cut, no slabs, no second launch; the result is the plain k-ordered fmaf chain per
// element (bit-identical to gemm_f32.hip's unsplit result: same chain, same order).
//
// One workgroup = 4 waves (2 x 2), wave tile (32 WT_M) x (32 WT_N) of 32x32 MFMA tiles, block tile BM = 64 WT_M, BN = 64 WT_N, k-tile 16.
// Staging: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a ring of 3 slots, two k-tiles ahead, counted vmcnt, one
// barrier per k-tile in the middle of a run of MFMAs; fragments of the next half-tile are read while the current one multiplies.
// Slot images as in gemm_f32.hip: NN A [16 k][BM m] (the global layout), B and TN-A [row][16 k] (64-byte rows) with the 16-byte chunk index
// XOR-swizzled by (row >> 2) & 3 (applied to the SOURCE address of the DMA). Operand shape of the MFMA: lane l supplies A[i = l & 31][k = l >> 5]
// and B[k = l >> 5][j = l & 31]; a lane's float4 along k feeds four consecutive MFMAs (any k permutation of a dot product is legal: the
// two half-waves take chunk 2 ks + h). NN: a lane's WT_M consecutive m feed WT_M different M-tiles, i.e. M-tile t holds rows WT_M i + t
// of the wave's block -- undone for free by the epilogue's float4 stores.
// Ragged M / N: DMA rows past the end are clamped to the last valid one, their results never stored. K % 16 != 0 (a multiple of 4 by the
// operator's precondition): one more k-tile after the pipelined loop, staged through registers with the missing k zero-filled.
#include "wg_internal.hpp"

#include <type_traits>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

constexpr int BK = 16;
constexpr int kThreads = 256;
constexpr int NRING = 3;

struct MidArgs {
    const float *a; uint32_t lda; uint64_t a_batch;
    const float *b; uint32_t ldb; uint64_t b_batch;
    float *c; uint32_t ldc; uint64_t c_batch;
    uint32_t M, N, K;
    uint32_t tiles_m, tiles_n;
    float alpha, beta;
    // split-K (k-split-wave tiles only; few tiles with a long K: 64 x 4096 x 4096): grid.y = nmats * nsplit, workgroup (z, s) covers
    // k in [s * k_per_split, ..) -- a multiple of 32 -- and writes the f32 slab (z, s) of `c` ([z][s][N][M], ldc = M); wg_splitk_reduce adds
    // them in ascending s and applies alpha / beta
    uint32_t nsplit, k_per_split;
};

__device__ __forceinline__ void store_c(float *p, float4 v, float alpha, float beta) {
    if (alpha != 1.f) { v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha; }
    if (beta != 0.f) {
        const float4 c = wg_ld_u(p);
        v.x = fmaf(beta, c.x, v.x); v.y = fmaf(beta, c.y, v.y); v.z = fmaf(beta, c.z, v.z); v.w = fmaf(beta, c.w, v.w);
    }
    wg_st_u(p, v);
}
__device__ __forceinline__ float comp(const float4 &v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }

// workgroup id -> tile: XCD x = id % 8 takes a contiguous range of the order index, which walks strips of 4 tile rows column by column
// (gemm_f32.hip's order: the workgroups an XCD runs at a time cover a compact patch of the output through its L2)
__device__ __forceinline__ void tile_of(uint32_t bid, uint32_t tiles_m, uint32_t tiles_n, uint32_t &tm, uint32_t &tn) {
    const uint32_t nwg = tiles_m * tiles_n;
    const uint32_t q = nwg / 8u, r = nwg % 8u, xcd = bid % 8u, local = bid / 8u;
    const uint32_t o = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + local;
    const uint32_t strip = o / (4u * tiles_n);
    const uint32_t within = o - strip * 4u * tiles_n;
    const uint32_t h = min(4u, tiles_m - 4u * strip);
    tn = within / h;
    tm = 4u * strip + (within - tn * h);
}

// one 1 KiB piece: LDS destination = M0 + 16 * lane. Nothing else in these kernels uses M0 (gfx950 DS instructions do not; checked on the ISA by
// tests/test_abi_and_host.py::test_f32_mid_kernels_own_m0), so it is simply overwritten.
__device__ __forceinline__ void dma16(uint32_t voff, const void *sbase, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst));
}
template <int N>
__device__ __forceinline__ void wait_dma_keep() { asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(N) : "memory"); }

template <bool TRANS_A, int WT_M, int WT_N>
__global__ __launch_bounds__(kThreads, (WT_M * WT_N == 4 ? 2 : 3)) void gemm_f32_mid_kernel(MidArgs g) {
    constexpr int BM = 64 * WT_M, BN = 64 * WT_N;
    constexpr int A_TILE = BM * BK, B_TILE = BN * BK, SLOT_FLOATS = A_TILE + B_TILE; // floats
    constexpr int NA = BM / 16, NB = BN / 16, PPW = (NA + NB) / 4;                    // 1 KiB DMA pieces per k-tile; per wave
    static_assert((NA + NB) % 4 == 0, "pieces must divide over the 4 waves");
    __shared__ __attribute__((aligned(16))) float smem[NRING * SLOT_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;

    uint32_t tm, tn;
    tile_of(blockIdx.x, g.tiles_m, g.tiles_n, tm, tn);
    const uint32_t m0 = tm * BM, n0 = tn * BN, z = blockIdx.y;
    const float *A = g.a + z * g.a_batch, *B = g.b + z * g.b_batch;
    float *C = g.c + z * g.c_batch;

    floatx16 acc[WT_M][WT_N];
#pragma unroll
    for (int t = 0; t < WT_M; ++t)
#pragma unroll
        for (int u = 0; u < WT_N; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

    // ---- DMA addressing: piece P = wave + 4 q; P < NA: piece P of A, else piece P - NA of B. Rows past the end: clamped ----
    uint32_t voff[PPW];
    const float *a_base, *b_base = B + (uint64_t)n0 * g.ldb;
    if constexpr (TRANS_A) a_base = A + (uint64_t)m0 * g.lda; else a_base = A + m0;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int P = wave + 4 * q;
        if (P < NA) {
            if constexpr (!TRANS_A) { // [16 k][BM m]: a piece = 256 floats = 256 / BM k-rows
                const uint32_t f = 256u * P + 4u * lane, krow = f / BM, m = min(f % BM, g.M - 4u - m0); // (M % 4 == 0)
                voff[q] = (krow * g.lda + m) * 4u;
            } else { // [BM rows][16 k]: a piece = 16 rows; lane -> row 16 P + (lane >> 2), chunk (lane & 3) ^ (lane >> 4)
                const uint32_t row = min(16u * P + (lane >> 2), g.M - 1u - m0);
                voff[q] = (row * g.lda + 4u * ((lane & 3) ^ (lane >> 4))) * 4u;
            }
        } else {
            const uint32_t row = min(16u * (P - NA) + (lane >> 2), g.N - 1u - n0);
            voff[q] = (row * g.ldb + 4u * ((lane & 3) ^ (lane >> 4))) * 4u;
        }
    }
    static_assert(NA % 4 == 0 && NB % 4 == 0, "a wave's piece q is an A piece for q < NA / 4, a B piece otherwise");
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)wave * 1024u); // + slot * SLOT bytes + q * 4 KiB
    // one piece of tile k0 -> ring slot `slot` (issued one at a time between MFMAs: a burst of them in front of the MFMAs is a bubble of its own)
    auto dma_piece = [&](uint32_t slot, uint32_t k0, int q) {
        const float *src;
        if (q < NA / 4) { if constexpr (TRANS_A) src = a_base + k0; else src = a_base + (uint64_t)k0 * g.lda; }
        else src = b_base + k0;
        dma16(voff[q], src, lds_wave + slot * (SLOT_FLOATS * 4) + (uint32_t)q * 4096u);
    };
    auto dma_tile = [&](uint32_t slot, uint32_t k0) {
#pragma unroll
        for (int q = 0; q < PPW; ++q) dma_piece(slot, k0, q);
    };
    static_assert(A_TILE * 4 == NA * 1024, "slot image");

    // ---- fragments: two register sets (by substep parity) ----
    typedef typename std::conditional<WT_M == 2, floatx2, float>::type a_nn_t;
    a_nn_t af_nn[2][4]; // NN: [set][step s]: WT_M consecutive m at k = 4 chunk + s
    float4 af_tn[2][WT_M], bf[2][WT_N];
    auto read_frags = [&](uint32_t slot, int ks, int set) {
        const float *As = smem + slot * SLOT_FLOATS;
        const float *Bs = As + A_TILE;
        const int chunk = 2 * ks + h; // this half-wave's 4 consecutive k within the 16-deep tile
        if constexpr (!TRANS_A) {
#pragma unroll
            for (int s = 0; s < 4; ++s) af_nn[set][s] = *reinterpret_cast<const a_nn_t *>(&As[(4 * chunk + s) * BM + wm * (BM / 2) + WT_M * i]);
        } else {
#pragma unroll
            for (int t = 0; t < WT_M; ++t) {
                const int mm = wm * (BM / 2) + 32 * t + i;
                af_tn[set][t] = *reinterpret_cast<const float4 *>(&As[mm * BK + 4 * (chunk ^ ((mm >> 2) & 3))]);
            }
        }
#pragma unroll
        for (int u = 0; u < WT_N; ++u) {
            const int nn = wn * (BN / 2) + 32 * u + i;
            bf[set][u] = *reinterpret_cast<const float4 *>(&Bs[nn * BK + 4 * (chunk ^ ((nn >> 2) & 3))]);
        }
    };
    auto a_of = [&](int set, int s, int t) -> float {
        if constexpr (TRANS_A) return comp(af_tn[set][t], s);
        else if constexpr (WT_M == 2) return af_nn[set][s][t];
        else return af_nn[set][s];
    };
    auto mfma_step = [&](int set, int s) {
#pragma unroll
        for (int t = 0; t < WT_M; ++t)
#pragma unroll
            for (int u = 0; u < WT_N; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_of(set, s, t), comp(bf[set][u], s), acc[t][u], 0, 0, 0);
    };
    // the four steps of a substep, one DMA piece behind each of the first PPW MFMAs
    auto mfma_substep_dma = [&](int set, uint32_t slot, uint32_t k0) {
        int n = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < WT_M; ++t)
#pragma unroll
                for (int u = 0; u < WT_N; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_of(set, s, t), comp(bf[set][u], s), acc[t][u], 0, 0, 0);
                    if (n < PPW) { dma_piece(slot, k0, n); __builtin_amdgcn_sched_barrier(0); }
                    ++n;
                }
    };
    static_assert(4 * WT_M * WT_N >= PPW, "a substep has an MFMA for every piece");

    const uint32_t nk = g.K / BK; // whole k-tiles (>= 1: launcher); a K % 16 remainder follows the pipelined loop
    auto next_slot = [](uint32_t x) { return x + 1 == NRING ? 0u : x + 1; };

    // The loop body has NO branches: every tile sends for "tile t + 2" and publishes "tile t + 1" -- past the end the cursor stays parked on the
    // last tile (its pieces land in slots nobody reads any more). A second copy of the MFMA sequence under a run-time condition made the
    // register allocator copy all accumulators every iteration (measured: 2048^3 126 -> 150 us).
    const uint32_t k_last = (nk - 1u) * BK;
    dma_tile(0, 0);
    dma_tile(1, min((uint32_t)BK, k_last));
    wait_dma_keep<PPW>();
    __syncthreads();
    read_frags(0, 0, 0);

    uint32_t cur = 0;
    for (uint32_t t = 0; t < nk; ++t) {
        const uint32_t nxt = next_slot(cur);
        // substep 0: multiply set 0; meanwhile read the second half-tile's fragments and send for tile t + 2, a piece behind each MFMA
        read_frags(cur, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_substep_dma(0, next_slot(nxt), min((t + 2u) * BK, k_last));
        // substep 1: multiply set 1; tile t + 1 has landed (sent a whole tile ago) -- publish it and read its first fragments
        mfma_step(1, 0);
        wait_dma_keep<PPW>();
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): this wave's fragment ds_reads have returned before another wave's DMA may overwrite the slot
        __builtin_amdgcn_s_barrier();
        read_frags(nxt, 0, 0);
#pragma unroll
        for (int s = 1; s < 4; ++s) mfma_step(1, s);
        cur = nxt;
    }
    wait_dma_keep<0>(); // the parked pieces: nothing may land once the ring is reused below

    // K % 16 != 0: the last, partial k-tile -- same slot image as the DMA writes (slot 0), missing k zero-filled, multiplied like any other.
    // Once per tile: not worth pipelining.
    if (nk * BK < g.K) {
        __syncthreads(); // every wave is done with the ring
        const uint32_t k0 = nk * BK;
        float *As = smem, *Bs = smem + A_TILE;
        for (int f = tid; f < A_TILE / 4; f += kThreads) {
            if constexpr (!TRANS_A) { // float4 along m at k-row f / (BM / 4)
                const uint32_t kr = (uint32_t)f / (BM / 4), m = min(4u * ((uint32_t)f % (BM / 4)), g.M - 4u - m0), k = k0 + kr;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < g.K) v = wg_ld_u(a_base + (uint64_t)k * g.lda + m);
                *reinterpret_cast<float4 *>(&As[kr * BM + 4 * (f % (BM / 4))]) = v;
            } else { // float4 along k: row f >> 2, chunk f & 3
                const int mm = f >> 2, ch = f & 3;
                const uint32_t k = k0 + 4u * ch, row = min((uint32_t)mm, g.M - 1u - m0);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < g.K) v = wg_ld_u(a_base + (uint64_t)row * g.lda + k);
                *reinterpret_cast<float4 *>(&As[mm * BK + 4 * (ch ^ ((mm >> 2) & 3))]) = v;
            }
        }
        for (int f = tid; f < B_TILE / 4; f += kThreads) {
            const int nn = f >> 2, ch = f & 3;
            const uint32_t k = k0 + 4u * ch, row = min((uint32_t)nn, g.N - 1u - n0);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < g.K) v = wg_ld_u(b_base + (uint64_t)row * g.ldb + k);
            *reinterpret_cast<float4 *>(&Bs[nn * BK + 4 * (ch ^ ((nn >> 2) & 3))]) = v;
        }
        __syncthreads();
        read_frags(0, 0, 0);
        read_frags(0, 1, 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) mfma_step(0, s);
#pragma unroll
        for (int s = 0; s < 4; ++s) mfma_step(1, s);
    }

    // ---- epilogue. C/D map of the 32x32 MFMA: lane l, register e -> row (e & 3) + 8 (e >> 2) + 4 (l >> 5), column l & 31 ----
#pragma unroll
    for (int u = 0; u < WT_N; ++u) {
        const uint32_t col = n0 + wn * (BN / 2) + 32 * u + i;
        if (col >= g.N) continue;
        float *cc = C + (uint64_t)col * g.ldc;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) { // e >> 2
            if constexpr (!TRANS_A && WT_M == 2) {
                // M-tile t holds rows 2 r + t of the wave's 64: (e & 3, t) enumerate 8 consecutive rows from 2 (8 gq + 4 h)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const uint32_t row = m0 + wm * 64 + 16 * gq + 8 * h + 4 * p;
                    if (row < g.M)
                        store_c(cc + row, make_float4(acc[0][u][4 * gq + 2 * p], acc[1][u][4 * gq + 2 * p], acc[0][u][4 * gq + 2 * p + 1], acc[1][u][4 * gq + 2 * p + 1]),
                                g.alpha, g.beta);
                }
            } else {
#pragma unroll
                for (int t = 0; t < WT_M; ++t) {
                    const uint32_t row = m0 + wm * (BM / 2) + 32 * t + 8 * gq + 4 * h;
                    if (row < g.M)
                        store_c(cc + row, make_float4(acc[t][u][4 * gq + 0], acc[t][u][4 * gq + 1], acc[t][u][4 * gq + 2], acc[t][u][4 * gq + 3]), g.alpha, g.beta);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Small tiles: 96 x 96, 96 x 64, 64 x 96, 64 x 64, 64 x 32, 32 x 64 -- the four waves of a workgroup SHARE one block tile and split K between them.
// Why: with 2 x 2 waves a 64 x 64 block leaves every wave ONE 32 x 32 accumulator, i.e. one dependent MFMA chain and a barrier every 8
// MFMAs (measured 63-70 % of the matrix cores' rate with one workgroup per CU: 1024^3 25 us against the vendor's 19.5). Here every wave
// holds the whole block (WT_M x WT_N independent accumulators) and multiplies ITS quarter of every k-tile: k-tile 32, wave w takes the 8 k of
// substep w (chunk 2 w + h of the half-wave h, steps s = 0..3) -- 4 WT_M WT_N MFMAs per wave between two barriers. The four partial blocks
// are added at the end through the LDS in a fixed order, (p0 + p2) + (p1 + p3): deterministic, but another association than the 2 x 2-wave
// tiles' single chain (both within the stated f32 bound, tests/test_gpu_ulp.py).
// Slot images (k-tile 32): NN A [32 k][BM m]; B and TN-A [row][32 k] = 128-byte rows, the 16-byte chunk c of a row at position
// c ^ ((row >> 1) & 7) (conflict-free for ds_read_b128 over 32 consecutive rows: MI355X_MICROARCH.md, LDS). Ring of 3 slots, tiles sent
// THREE ahead right after the barrier that frees their slot (every fragment of a tile is in registers before that barrier).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BK2 = 32;

template <bool TRANS_A, int WT_M, int WT_N>
__global__ __launch_bounds__(kThreads, 2) void gemm_f32_mid_kw_kernel(MidArgs g) {
    constexpr int BM = 32 * WT_M, BN = 32 * WT_N, NT = WT_M * WT_N;
    constexpr int A_TILE = BM * BK2, B_TILE = BN * BK2, SLOT_FLOATS = A_TILE + B_TILE;
    constexpr int NA = BM / 8, NB = BN / 8, PPW = (NA + NB) / 4; // 1 KiB pieces per k-tile (A, B); per wave
    static_assert((NA + NB) % 4 == 0, "pieces must divide over the 4 waves");
    constexpr int RING_FLOATS = NRING * SLOT_FLOATS, RED_FLOATS = 2 * NT * 16 * 64; // the reduction's first phase: two waves' accumulators
    __shared__ __attribute__((aligned(16))) float smem[RING_FLOATS > RED_FLOATS ? RING_FLOATS : RED_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;

    uint32_t tm, tn;
    tile_of(blockIdx.x, g.tiles_m, g.tiles_n, tm, tn);
    const uint32_t m0 = tm * BM, n0 = tn * BN, z = blockIdx.y / g.nsplit, split = blockIdx.y % g.nsplit;
    const uint32_t k_begin = split * g.k_per_split, K = min(g.K - k_begin, g.k_per_split); // (workgroup-uniform; K >= 32: launcher)
    const float *A = g.a + z * g.a_batch + (TRANS_A ? (uint64_t)k_begin : (uint64_t)k_begin * g.lda), *B = g.b + z * g.b_batch + k_begin;
    float *C = g.c + ((uint64_t)z * g.nsplit + split) * g.c_batch;

    floatx16 acc[WT_M][WT_N];
#pragma unroll
    for (int t = 0; t < WT_M; ++t)
#pragma unroll
        for (int u = 0; u < WT_N; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

    // ---- DMA addressing: piece P = wave + 4 q; P < NA: piece P of A, else piece P - NA of B; rows past the end clamped ----
    uint32_t voff[PPW];
    const float *a_base, *b_base = B + (uint64_t)n0 * g.ldb;
    if constexpr (TRANS_A) a_base = A + (uint64_t)m0 * g.lda; else a_base = A + m0;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int P = wave + 4 * q;
        if (P < NA) {
            if constexpr (!TRANS_A) { // [32 k][BM m]: a piece = 256 floats = 256 / BM k-rows
                const uint32_t f = 256u * P + 4u * lane, krow = f / BM, m = min(f % BM, g.M - 4u - m0);
                voff[q] = (krow * g.lda + m) * 4u;
            } else { // [BM rows][32 k]: a piece = 8 rows of 128 bytes; lane -> row 8 P + (lane >> 3), position lane & 7
                const uint32_t r = 8u * P + (lane >> 3), row = min(r, g.M - 1u - m0);
                voff[q] = (row * g.lda + 4u * ((lane & 7u) ^ ((r >> 1) & 7u))) * 4u;
            }
        } else {
            const uint32_t r = 8u * (P - NA) + (lane >> 3), row = min(r, g.N - 1u - n0);
            voff[q] = (row * g.ldb + 4u * ((lane & 7u) ^ ((r >> 1) & 7u))) * 4u;
        }
    }
    static_assert(A_TILE * 4 == NA * 1024, "slot image");
    static_assert(NA % 4 == 0 && NB % 4 == 0, "a wave's piece q is an A piece for q < NA / 4, a B piece otherwise");
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)wave * 1024u);
    auto dma_piece = [&](uint32_t slot, uint32_t k0, int q) {
        const float *src;
        if (q < NA / 4) { if constexpr (TRANS_A) src = a_base + k0; else src = a_base + (uint64_t)k0 * g.lda; }
        else src = b_base + k0;
        dma16(voff[q], src, lds_wave + slot * (SLOT_FLOATS * 4) + (uint32_t)q * 4096u);
    };
    auto dma_tile = [&](uint32_t slot, uint32_t k0) {
#pragma unroll
        for (int q = 0; q < PPW; ++q) dma_piece(slot, k0, q);
    };

    // ---- this wave's fragments of one k-tile: chunk 2 wave + h, steps s = 0..3; two register sets (tile parity) ----
    struct a_nn_t { float v[WT_M]; }; // WT_M consecutive m (one per M-tile)
    a_nn_t af_nn[2][4];
    float4 af_tn[2][WT_M], bf[2][WT_N];
    const int chunk = 2 * wave + h;
    auto read_frags = [&](uint32_t slot, int set) {
        const float *As = smem + slot * SLOT_FLOATS;
        const float *Bs = As + A_TILE;
        if constexpr (!TRANS_A) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < WT_M; ++t) af_nn[set][s].v[t] = As[(4 * chunk + s) * BM + WT_M * i + t];
        } else {
#pragma unroll
            for (int t = 0; t < WT_M; ++t) {
                const int mm = 32 * t + i;
                af_tn[set][t] = *reinterpret_cast<const float4 *>(&As[mm * BK2 + 4 * (chunk ^ ((mm >> 1) & 7))]);
            }
        }
#pragma unroll
        for (int u = 0; u < WT_N; ++u) {
            const int nn = 32 * u + i;
            bf[set][u] = *reinterpret_cast<const float4 *>(&Bs[nn * BK2 + 4 * (chunk ^ ((nn >> 1) & 7))]);
        }
    };
    auto a_of = [&](int set, int s, int t) -> float {
        if constexpr (TRANS_A) return comp(af_tn[set][t], s);
        else return af_nn[set][s].v[t];
    };
    auto mfma_step = [&](int set, int s) {
#pragma unroll
        for (int t = 0; t < WT_M; ++t)
#pragma unroll
            for (int u = 0; u < WT_N; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_of(set, s, t), comp(bf[set][u], s), acc[t][u], 0, 0, 0);
    };

    const uint32_t nk = K / BK2; // whole k-tiles (>= 1: launcher); a K % 32 remainder follows the pipelined loop
    auto next_slot = [](uint32_t x) { return x + 1 == NRING ? 0u : x + 1; };
    static_assert(2 * NT >= PPW, "the second half of a tile has an MFMA for every piece");
    // Branch-free body (see the 2 x 2-wave kernel): every tile publishes "tile t + 1" and sends for "tile t + 3" into its own slot -- all of its
    // fragments are in registers on every wave once the barrier is passed --, one piece behind each MFMA; past the end the cursor is parked.
    const uint32_t k_last = (nk - 1u) * BK2;
    auto tile_body = [&](uint32_t t, uint32_t cur, auto set_c) {
        constexpr int set = decltype(set_c)::value;
        mfma_step(set, 0);
        mfma_step(set, 1);
        wait_dma_keep<PPW>(); // tile t + 1 (sent two barriers ago) has landed; tile t + 2's pieces may stay in flight
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): "all fragments are in registers" made true before the slot is handed to the DMA
        __builtin_amdgcn_s_barrier();
        read_frags(next_slot(cur), set ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t k_dma = min((t + 3u) * BK2, k_last);
        int n = 0;
#pragma unroll
        for (int s = 2; s < 4; ++s)
#pragma unroll
            for (int tt = 0; tt < WT_M; ++tt)
#pragma unroll
                for (int u = 0; u < WT_N; ++u) {
                    acc[tt][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_of(set, s, tt), comp(bf[set][u], s), acc[tt][u], 0, 0, 0);
                    if (n < PPW) { dma_piece(cur, k_dma, n); __builtin_amdgcn_sched_barrier(0); }
                    ++n;
                }
    };

    for (uint32_t t = 0; t < 3; ++t) dma_tile(t, min(t * BK2, k_last));
    wait_dma_keep<2 * PPW>();
    __syncthreads();
    read_frags(0, 0);
    uint32_t cur = 0, t = 0;
    for (; t + 1 < nk; t += 2) {
        tile_body(t, cur, std::integral_constant<int, 0>{});
        cur = next_slot(cur);
        tile_body(t + 1, cur, std::integral_constant<int, 1>{});
        cur = next_slot(cur);
    }
    if (t < nk) tile_body(t, cur, std::integral_constant<int, 0>{});
    wait_dma_keep<0>(); // the parked pieces: nothing may land once the ring is reused below

    // K % 32 != 0 (a multiple of 4): the last, partial k-tile through registers into slot 0, missing k zero-filled, multiplied like any other
    if (nk * BK2 < K) {
        __syncthreads(); // every wave is done with the ring
        const uint32_t k0 = nk * BK2;
        float *As = smem, *Bs = smem + A_TILE;
        for (int f = tid; f < A_TILE / 4; f += kThreads) {
            if constexpr (!TRANS_A) {
                const uint32_t kr = (uint32_t)f / (BM / 4), m = min(4u * ((uint32_t)f % (BM / 4)), g.M - 4u - m0), k = k0 + kr;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < K) v = wg_ld_u(a_base + (uint64_t)k * g.lda + m);
                *reinterpret_cast<float4 *>(&As[kr * BM + 4 * (f % (BM / 4))]) = v;
            } else {
                const int mm = f >> 3, ch = f & 7;
                const uint32_t k = k0 + 4u * ch, row = min((uint32_t)mm, g.M - 1u - m0);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < K) v = wg_ld_u(a_base + (uint64_t)row * g.lda + k);
                *reinterpret_cast<float4 *>(&As[mm * BK2 + 4 * (ch ^ ((mm >> 1) & 7))]) = v;
            }
        }
        for (int f = tid; f < B_TILE / 4; f += kThreads) {
            const int nn = f >> 3, ch = f & 7;
            const uint32_t k = k0 + 4u * ch, row = min((uint32_t)nn, g.N - 1u - n0);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < K) v = wg_ld_u(b_base + (uint64_t)row * g.ldb + k);
            *reinterpret_cast<float4 *>(&Bs[nn * BK2 + 4 * (ch ^ ((nn >> 1) & 7))]) = v;
        }
        __syncthreads();
        read_frags(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) mfma_step(0, s);
    }

    // ---- the four partial blocks: (p0 + p2) + (p1 + p3) through the LDS; waves 0 / 1 end up with the columns u = 0 / 1 (WT_N = 2), or
    // wave 0 with everything ----
    __syncthreads(); // the ring is no longer read
    float4 *red = reinterpret_cast<float4 *>(smem);
    auto put = [&](int slab, int t, int u) { // this wave's accumulator (t, u) -> slab (4 KiB each)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            red[((slab * NT + t * WT_N + u) * 4 + q) * 64 + lane] = make_float4(acc[t][u][4 * q], acc[t][u][4 * q + 1], acc[t][u][4 * q + 2], acc[t][u][4 * q + 3]);
    };
    auto add = [&](int slab, int t, int u) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = red[((slab * NT + t * WT_N + u) * 4 + q) * 64 + lane];
            acc[t][u][4 * q] += v.x; acc[t][u][4 * q + 1] += v.y; acc[t][u][4 * q + 2] += v.z; acc[t][u][4 * q + 3] += v.w;
        }
    };
    if (wave >= 2) {
#pragma unroll
        for (int t = 0; t < WT_M; ++t)
#pragma unroll
            for (int u = 0; u < WT_N; ++u) put(wave - 2, t, u);
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int t = 0; t < WT_M; ++t)
#pragma unroll
            for (int u = 0; u < WT_N; ++u) add(wave, t, u); // p0 + p2, p1 + p3
    }
    __syncthreads();
    constexpr int U_SPLIT = WT_N == 2 ? 1 : 0; // 1: wave 1 keeps the columns u = 1 and hands u = 0 over, wave 0 the other way round (else: all to wave 0)
    if (wave < 2) {
#pragma unroll
        for (int t = 0; t < WT_M; ++t)
#pragma unroll
            for (int u = 0; u < WT_N; ++u)
                if (U_SPLIT ? (u != wave) : (wave == 1)) put(wave, t, u);
    }
    __syncthreads();
    if (wave >= 2 || (!U_SPLIT && wave == 1)) return;
#pragma unroll
    for (int u = 0; u < WT_N; ++u) {
        if (U_SPLIT && u != wave) continue;
#pragma unroll
        for (int t = 0; t < WT_M; ++t) {
            if (wave == 0) add(1, t, u); // (p0 + p2) + (p1 + p3)
            else { // the same order for wave 1's columns: what wave 0 handed over is the LEFT operand
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = red[((0 * NT + t * WT_N + u) * 4 + q) * 64 + lane];
                    acc[t][u][4 * q] = v.x + acc[t][u][4 * q]; acc[t][u][4 * q + 1] = v.y + acc[t][u][4 * q + 1];
                    acc[t][u][4 * q + 2] = v.z + acc[t][u][4 * q + 2]; acc[t][u][4 * q + 3] = v.w + acc[t][u][4 * q + 3];
                }
            }
        }
        const uint32_t col = n0 + 32 * u + i;
        if (col >= g.N) continue;
        float *cc = C + (uint64_t)col * g.ldc;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            if constexpr (!TRANS_A && WT_M > 1) {
                // M-tile t holds rows WT_M r + t of the block: (e & 3, t) enumerate 4 WT_M consecutive rows from WT_M (8 gq + 4 h)
#pragma unroll
                for (int p = 0; p < WT_M; ++p) {
                    const uint32_t row = m0 + WT_M * (8 * gq + 4 * h) + 4 * p;
                    auto val = [&](int n) { return acc[n % WT_M][u][4 * gq + n / WT_M]; };
                    if (row < g.M) store_c(cc + row, make_float4(val(4 * p), val(4 * p + 1), val(4 * p + 2), val(4 * p + 3)), g.alpha, g.beta);
                }
            } else {
#pragma unroll
                for (int t = 0; t < WT_M; ++t) {
                    const uint32_t row = m0 + 32 * t + 8 * gq + 4 * h;
                    if (row < g.M)
                        store_c(cc + row, make_float4(acc[t][u][4 * gq + 0], acc[t][u][4 * gq + 1], acc[t][u][4 * gq + 2], acc[t][u][4 * gq + 3]), g.alpha, g.beta);
                }
            }
        }
    }
}

template <int WT_M, int WT_N>
int launch_kw(wg_ctx *ctx, bool trans, uint32_t nmats, const MidArgs &g) {
    const dim3 grid(g.tiles_m * g.tiles_n, nmats * g.nsplit), block(kThreads);
    if (trans) hipLaunchKernelGGL((gemm_f32_mid_kw_kernel<true, WT_M, WT_N>), grid, block, 0, ctx->stream, g);
    else hipLaunchKernelGGL((gemm_f32_mid_kw_kernel<false, WT_M, WT_N>), grid, block, 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

template <int WT_M, int WT_N>
int launch(wg_ctx *ctx, bool trans, uint32_t nmats, const MidArgs &g) {
    const dim3 grid(g.tiles_m * g.tiles_n, nmats), block(kThreads);
    if (trans) hipLaunchKernelGGL((gemm_f32_mid_kernel<true, WT_M, WT_N>), grid, block, 0, ctx->stream, g);
    else hipLaunchKernelGGL((gemm_f32_mid_kernel<false, WT_M, WT_N>), grid, block, 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

} // namespace

bool wgk_gemm_f32_mid_ok(uint32_t M, uint32_t N, uint32_t K, uint32_t nmats, wgk_mat m1, wgk_mat m2) {
    // at least one whole k-tile of either family; 32-bit byte offsets inside a tile's 128 rows / 16 k-rows; grid.y
    return K >= 32 && K % 4 == 0 && M >= 4 && N >= 4 && nmats <= 65535 && (uint64_t)m1.ld * 128u * 4u < (1ull << 31) && (uint64_t)m2.ld * 128u * 4u < (1ull << 31);
}

int wgk_gemm_f32_mid(wg_ctx *ctx, bool trans, int bm, int bn, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats, float *out, uint32_t out_ld, uint64_t out_batch,
                     wgk_mat m1, wgk_mat m2, float alpha, float beta, uint32_t nsplit) {
    const bool kw = !(bm == 128 || bn == 128);
    if (nsplit > 1 && !kw) nsplit = 1; // (the 2 x 2-wave tiles have no split form)
    uint32_t kps = K;
    for (uint32_t want = nsplit; nsplit > 1; --want) { // whole k-tiles per split, no empty split, at least one whole k-tile in the last one
        if (want <= 1) { nsplit = 1; kps = K; break; }
        kps = (((K + 31u) / 32u + want - 1u) / want) * 32u;
        const uint32_t n = (K + kps - 1u) / kps;
        if (n > 1 && K - (n - 1u) * kps >= 32u) { nsplit = n; break; }
    }
    if ((uint64_t)nmats * nsplit > 65535u) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: nmats * splits exceeds 65535");
    MidArgs g;
    g.a = (const float *)m1.ptr; g.lda = m1.ld; g.a_batch = m1.batch;
    g.b = (const float *)m2.ptr; g.ldb = m2.ld; g.b_batch = m2.batch;
    g.c = out; g.ldc = out_ld; g.c_batch = out_batch;
    g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta;
    g.nsplit = nsplit > 1 ? nsplit : 1u; g.k_per_split = kps;
    float *part = nullptr;
    if (g.nsplit > 1) { // raw partial sums into f32 slabs; alpha / beta are applied by the ordered reduce
        void *ws = nullptr;
        if (int rc = wg_ctx_workspace(ctx, (size_t)g.nsplit * M * N * nmats * sizeof(float), &ws)) return rc;
        part = (float *)ws;
        g.c = part; g.ldc = M; g.c_batch = (uint64_t)M * N;
        g.alpha = 1.f; g.beta = 0.f;
    }
    g.tiles_m = (M + (uint32_t)bm - 1) / (uint32_t)bm;
    g.tiles_n = (N + (uint32_t)bn - 1) / (uint32_t)bn;
    if ((uint64_t)g.tiles_m * g.tiles_n > 0x7fffffffull) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: too many tiles");
    int rc = WG_ERR_INVALID_ARG;
    if (bm == 128 && bn == 128) rc = launch<2, 2>(ctx, trans, nmats, g);
    else if (bm == 128 && bn == 64) rc = launch<2, 1>(ctx, trans, nmats, g);
    else if (bm == 64 && bn == 128) rc = launch<1, 2>(ctx, trans, nmats, g);
    else if (bm == 64 && bn == 64) rc = launch_kw<2, 2>(ctx, trans, nmats, g);
    else if (bm == 64 && bn == 32) rc = launch_kw<2, 1>(ctx, trans, nmats, g);
    else if (bm == 32 && bn == 64) rc = launch_kw<1, 2>(ctx, trans, nmats, g);
    else if (bm == 96 && bn == 96) rc = launch_kw<3, 3>(ctx, trans, nmats, g);
    else if (bm == 96 && bn == 64) rc = launch_kw<3, 2>(ctx, trans, nmats, g);
    else if (bm == 64 && bn == 96) rc = launch_kw<2, 3>(ctx, trans, nmats, g);
    else return wg_set_error(WG_ERR_INVALID_ARG, "Gemm: no %d x %d f32 tile", bm, bn);
    if (rc != WG_OK) return rc;
    if (g.nsplit > 1) return wg_splitk_reduce(ctx, part, g.nsplit, M, N, nmats, WG_F32, out, out_ld, out_batch, alpha, beta);
    return WG_OK;
}
