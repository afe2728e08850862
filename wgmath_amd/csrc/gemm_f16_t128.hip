// f16 Gemm, mid-size variants: 128 x 128 and 256 x 128 block tiles, two workgroups per CU.
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
//     ds_read_b64_tr_b16 for column-major A -- since round 6 without v_permlane16_swap: the unit-parity interleave of gemm_f16.hip "NN A"),
//     and the same row permutation inside a wave tile so that a lane's accumulators of a tile pair are 8 consecutive rows of C (16-byte stores);
//   * tile order: 4-tile-row strips, contiguous ranges per XCD (tile_strips, gemm_f16_common.hpp), so an XCD's L2 sees a compact
//     patch of the output.
// M0 is written by the inline asm without save/restore, as in gemm_f16.hip (tests/test_abi_and_host.py checks the ISA).
// Bound: LDS bandwidth (per CU and half-step pair: 64 KiB of fragment reads + 32 KiB of DMA writes against 512 MFMA cycles per
// SIMD) and L2 -> CU bandwidth; see DESIGN.md section 3.
//
// TM = 256 (round 4): the same kernel with 2 x 2 waves of 128 x 64 = 8 x 4 MFMA tiles (128 accumulator registers: still two workgroups per CU). What it is
// for: products whose K is short against the big kernel's per-tile costs. gemm_f16.hip owns a CU per workgroup, so a tile's prologue (2.8 us), store
// burst (6.3 us) and re-dispatch are serial with its main loop (21 us at K = 1024): a third of the time no MFMA runs. Two co-resident workgroups of half
// the tile cover each other's prologue and epilogue. Per wave and half-step: 32 MFMAs between two barriers (128 x 128: 16), 12 KiB of fragment reads (8),
// 6 DMA pieces (4); a half-stage is A 256 x 32 + B 32 x 128 = 24 KiB and the ring has 3 slots (72 KiB), i.e. ONE half-stage of lead beyond the one being read --
// the other workgroup covers the rest.
#include "gemm_f16_common.hpp"

namespace wgf16 {
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef uintx4 __attribute__((aligned(2))) uintx4_u;

template <class F, int... I>
__device__ __forceinline__ void t_static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void t_static_for(F &&f) { t_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

#ifndef WG_T128_ABLATE
#define WG_T128_ABLATE 0 // timing experiments only (results are garbage): 1 = no MFMA / fragment reads, 2 = no DMA, 4 = no barriers
#endif
constexpr int TN = 128;
// per block height TM (128 / 256): APW A pieces per wave and half-stage, the wave's region of a slot [A pieces APW w .. | B pieces 2 w, 2 w + 1], the slot, the ring
template <int TM> struct TCfg {
    static constexpr int APW = TM / 64, PPW = APW + 2, RB = PPW * 1024, SLOT = 4 * RB, RING = TM == 128 ? 5 : 3, MT = TM / 32, MP = MT / 2;
    static_assert(TM == 128 || TM == 256, "block heights built");
};
constexpr uint32_t T_BIAS = 3072;   // see M16_BIAS in gemm_f16.hip

__device__ __forceinline__ void t_set_m0(uint32_t lds_dst) { asm volatile("s_mov_b32 m0, %0" ::"s"(lds_dst)); }
template <int IMM>
__device__ __forceinline__ void t_dma(uint32_t voff, const void *sbase) {
    if (WG_T128_ABLATE & 2) return;
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%c2" ::"v"(voff), "s"(sbase), "i"(IMM));
}

// op(A)'s pieces when A is read once and cannot stay in the Infinity Cache (GemmArgs::a_nt: few columns on a tall matrix): the non-temporal hint, as in the few-column
// f32 kernel (gemm_f32_skinny.hip tr_dma_streamed; profiles/r05_skinny_nt_ab.txt)
template <int IMM>
__device__ __forceinline__ void t_dma_streamed(uint32_t voff, const void *sbase, uint32_t nt) {
    if (WG_T128_ABLATE & 2) return;
    if (nt) asm volatile("global_load_lds_dwordx4 %0, %1 offset:%c2 nt" ::"v"(voff), "s"(sbase), "i"(IMM));
    else asm volatile("global_load_lds_dwordx4 %0, %1 offset:%c2" ::"v"(voff), "s"(sbase), "i"(IMM));
}

template <int N, int STEP>
__device__ __forceinline__ void t_wait_keep_pieces(int n) { // s_waitcnt's count is an immediate: a run-time count (a multiple of STEP below STEP * N) picks its instruction
    if constexpr (N > 0) {
        if (n == STEP * (N - 1)) wait_dma_keep<STEP * (N - 1)>();
        else t_wait_keep_pieces<N - 1, STEP>(n);
    }
}

// B_NC (round 6; Gemm only, K % 64 == 0): m2 contiguous along N -- element (k, n) at b + n + k * ldb -- i.e. the row-major GemmTr in column-major terms (gemm_f16_nt.hip
// is the 256 x 256 form). B then takes column-major A's path: pieces = k-quads of 256-byte [4 k][32 n] blocks, swap-free transposing reads, N tile 2 p' + tb of the wave
// holding column 32 p' + 8 a + 4 (tb ^ (a & 1)) + e for lane i16 = 4 a + e (the epilogue's column index).
template <bool TRANS_A, int TM, bool B_NC = false>
__global__ __launch_bounds__(256, 2) void gemm_f16_t128_kernel(GemmArgs g) {
    static_assert(!(B_NC && TRANS_A) && (!B_NC || WG_NN_NOSWAP), "n-contiguous m2: Gemm only, on the swap-free read path");
    using Cfg = TCfg<TM>;
    constexpr int APW = Cfg::APW, PPW = Cfg::PPW, RB = Cfg::RB, T_SLOT = Cfg::SLOT, T_RING = Cfg::RING, MT = Cfg::MT, MP = Cfg::MP;
    __shared__ __attribute__((aligned(16))) char smem[T_RING * T_SLOT];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i16 = lane & 15, kg = lane >> 4;
    const int aq = i16 >> 2, bb = i16 & 3;
    const int gq = (4 - aq) & 3; // G(aq), G = {0,3,2,1}

    uint32_t tm, tn;
    tile_strips(blockIdx.x, g.tiles_m, g.tiles_n, tm, tn);
    const uint32_t m0 = tm * TM, n0 = tn * TN;
    const uint32_t z = blockIdx.y / g.nsplit, split = blockIdx.y % g.nsplit;
    const uint32_t k_begin = split * g.k_per_split;
    // K remainder (K % 64, a multiple of 8; the last split owns it): multiplied FIRST, as two zero-padded half-stages written to LDS slots 0 and 1
    // by ordinary loads + stores in exactly the image the DMA pieces would have left (m16_tile in gemm_f16.hip does the same); the DMA stream
    // then starts at half-stage 2 = the first 32 k of this split (bases moved back by two half-stages below).
    uint32_t K_loc = split + 1u == g.nsplit ? g.K - k_begin : g.k_per_split;
    const uint32_t rem = K_loc & 63u;
    K_loc -= rem;
    const uint32_t rem_dk = K_loc; // the remainder's first k, relative to k_begin
    const _Float16 *A = g.a + z * g.a_batch + (TRANS_A ? (uint64_t)k_begin : (uint64_t)k_begin * g.lda);
    const _Float16 *B = g.b + z * g.b_batch + (B_NC ? (uint64_t)k_begin * g.ldb : (uint64_t)k_begin);

    // ---- DMA addressing: per half-stage this wave stages pieces P = 2 wave + q (q = 0, 1) of A and of B; a piece = 1 KiB of LDS
    // (64 lanes x 16 bytes). k-contiguous operands (B; op(A) for TN): piece P = rows 16P..16P+15 of 64 bytes (32 k), lane -> row
    // 16P + (lane>>2), position lane&3 holds the logical chunk (lane&3) ^ G(key(row)), key = (row>>2)&3 for B (rows read in natural
    // order) and (row>>3)&3 for op(A) (rows read permuted). Column-major A: piece P = k-quad kq = P, blocks [mblk = lane>>4] of
    // [4 k][32 m]: k row (lane>>2)&3, 16-byte unit lane&3. Rows past the end of a ragged tile are clamped (results never stored).
    // (TM = 256: the wave stages A pieces 4 wave + q, q < 4; a column-major k-quad is 2 pieces -- m halves -- so P = 2 kq + half. Its two B pieces lie beyond the
    // instruction offset's 4095 bytes from the region's start: a second M0 value, B_IMM0 = 0; TM = 128: one M0, B_IMM0 = 2048, as before)
    constexpr int B_IMM0 = APW == 2 ? 2048 : 0;
    uint32_t a_voff[APW], b_voff[2];
    const _Float16 *a_base, *b_base = B_NC ? B + n0 : B + (uint64_t)n0 * g.ldb;
    if constexpr (TRANS_A) a_base = A + (uint64_t)m0 * g.lda; else a_base = A + m0;
#pragma unroll
    for (int q = 0; q < APW; ++q) {
        const uint32_t P = (uint32_t)APW * wave + q;
        if constexpr (TRANS_A) {
            const uint32_t row = 16u * P + (lane >> 2);
            const uint32_t ra = min(row, g.M - 1u - m0);
            a_voff[q] = (ra * g.lda + 8u * ((lane & 3u) ^ ((4u - ((row >> 3) & 3u)) & 3u))) * 2u + (T_BIAS - 1024u * q);
        } else {
            const uint32_t kq = APW == 2 ? P : P >> 1, mh = APW == 2 ? 0u : (P & 1u);
            const uint32_t k = 4u * kq + ((lane >> 2) & 3u);
            // (swap-free form, gemm_f16.hip "NN A": the piece's k-group within the half-stage is kq >> 1 = `wave`; odd ones swap neighbouring 16-byte units)
            const uint32_t unit = WG_NN_NOSWAP ? (lane & 3u) ^ ((uint32_t)wave & 1u) : (lane & 3u);
            const uint32_t m = min(128u * mh + 32u * (lane >> 4) + 8u * unit, g.M - 8u - m0); // M % 8 == 0
            a_voff[q] = (k * g.lda + m) * 2u + (T_BIAS - 1024u * q);
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if constexpr (B_NC) { // piece 2 wave + q = k-quad kq of the half-stage (its k-group: `wave`): blocks [n/32 = lane >> 4], k row (lane >> 2) & 3, unit (lane & 3) ^ (wave & 1)
            const uint32_t k = 4u * (2u * wave + q) + ((lane >> 2) & 3u);
            const uint32_t n = min(32u * (lane >> 4) + 8u * ((lane & 3u) ^ ((uint32_t)wave & 1u)), g.N - 8u - n0); // N % 8 == 0 (launcher)
            b_voff[q] = (k * g.ldb + n) * 2u + (T_BIAS - (uint32_t)B_IMM0 - 1024u * q);
            continue;
        }
        const uint32_t row = 16u * (2u * wave + q) + (lane >> 2);
        const uint32_t rb = min(row, g.N - 1u - n0);
        b_voff[q] = (rb * g.ldb + 8u * ((lane & 3u) ^ ((4u - ((row >> 2) & 3u)) & 3u))) * 2u + (T_BIAS - (uint32_t)B_IMM0 - 1024u * q);
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(WG_AS3 char *)smem;
    const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(lds_base + wave * RB);
    const uint64_t a_step = TRANS_A ? 64u : (uint64_t)64u * g.lda; // bytes per half-stage (32 k)
    const char *ga0 = (const char *)a_base - T_BIAS - (rem ? 2u * a_step : 0u), *gb0 = (const char *)b_base - T_BIAS - (rem ? 128u : 0u);
    const uint32_t a_nt = g.a_nt; // (workgroup-uniform)
    const uint64_t b_step = B_NC ? (uint64_t)64u * g.ldb : 64u; // bytes per half-stage of B
    auto issue = [&](uint32_t H, uint32_t slot_off) { // this wave's PPW pieces of half-stage H into the slot at byte offset slot_off
        const char *ga = ga0 + H * a_step, *gb = gb0 + (uint64_t)H * b_step;
        t_set_m0(lds_wave + slot_off);
        asm volatile("s_nop 0");
        t_dma_streamed<0>(a_voff[0], ga, a_nt); t_dma_streamed<1024>(a_voff[1], ga, a_nt);
        if constexpr (APW == 4) {
            t_dma_streamed<2048>(a_voff[2], ga, a_nt); t_dma_streamed<3072>(a_voff[3], ga, a_nt);
            t_set_m0(lds_wave + slot_off + 4096u);
            asm volatile("s_nop 0");
        }
        t_dma<B_IMM0>(b_voff[0], gb); t_dma<B_IMM0 + 1024>(b_voff[1], gb);
    };

    // ---- per-lane LDS read offsets within a slot ----
    // B tile u: row 64 wn + 16 u + i16 = piece 4 wn + u, row i16 of it -> (2 wn + (u>>1)) * 4096 + 2048 + (u&1) * 1024 + i16 * 64 + pos * 16
    const uint32_t pos = (uint32_t)(kg ^ gq);
    const uint32_t b_off = 2u * wn * (uint32_t)RB + (uint32_t)APW * 1024u + (uint32_t)i16 * 64u + pos * 16u;
    // A, TN: MFMA tile t = 2 p + tb, MFMA row i16 <-> tile row 64 wm + 32 p + 8 aq + 4 tb + bb = piece 4 wm + 2 p + (aq>>1), row 8 (aq&1) + 4 tb + bb
    // A, NN: transpose read i = 2 h + ins of pair p: k-quad kq = 4 (kg>>1) + 2 ins + h (piece kq), block mblk = 2 wm + p, unit i16, half kg&1
    // (TM = 256, TN: the wave's rows (TM / 2) wm + 32 p + ..: piece 8 wm + 2 p + (aq>>1) = region 2 wm + (p>>1), sub-piece 2 (p&1) + (aq>>1);
    //  NN: k-quad kq = 2 pieces [m half 0 | m half 1], the wave's half is wm, block p of it: region kq >> 1, + ((kq & 1) * 2 + wm) KiB + p * 256)
    uint32_t a_off[2];
    if constexpr (TRANS_A) {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
            a_off[tb] = 2u * wm * (uint32_t)RB + (uint32_t)(aq >> 1) * 1024u + (uint32_t)(8 * (aq & 1) + 4 * tb + bb) * 64u + pos * 16u;
    } else if constexpr (WG_NN_NOSWAP) {
        // lane row kg reads k-group kg = region kg of the slot; lane 4 krow + a: k row krow, unit a (at position a ^ (kg & 1)), 8-byte half tb ^ (a & 1) for tile tb
        const uint32_t a = (uint32_t)i16 & 3u, krow = (uint32_t)i16 >> 2;
        const uint32_t common = (uint32_t)kg * (uint32_t)RB + (APW == 2 ? (2u * wm) * 256u : (uint32_t)wm * 1024u) + krow * 64u + (a ^ ((uint32_t)kg & 1u)) * 16u;
        a_off[0] = common + (a & 1u) * 8u;
        a_off[1] = common + ((a & 1u) ^ 1u) * 8u;
    } else {
        a_off[0] = (uint32_t)(2 * (kg >> 1)) * (uint32_t)RB + (APW == 2 ? (2u * wm) * 256u : (uint32_t)wm * 1024u) + (uint32_t)i16 * 16u + (uint32_t)(kg & 1) * 8u;
        a_off[1] = 0;
    }
    const bool odd_row = !TRANS_A && WG_NN_NOSWAP && (kg & 1); // these lanes hold the tiles of a pair in exchanged order (rows + 4..7 in tile 2 p)
    uint32_t bn_off[2] = { 0, 0 }; // B_NC: lane row kg reads k-group kg = region kg, behind the region's A pieces; blocks n/32 = 2 wn + p'; even / odd N tiles of a pair
    if constexpr (B_NC) {
        const uint32_t a = (uint32_t)i16 & 3u, krow = (uint32_t)i16 >> 2;
        const uint32_t common = (uint32_t)kg * (uint32_t)RB + (uint32_t)APW * 1024u + (2u * wn) * 256u + krow * 64u + (a ^ ((uint32_t)kg & 1u)) * 16u;
        bn_off[0] = common + (a & 1u) * 8u;
        bn_off[1] = common + ((a & 1u) ^ 1u) * 8u;
    }
    constexpr auto a_pair_off = [](int p) { return APW == 2 ? p * RB : (p >> 1) * RB + (p & 1) * 2048; }; // TN: where pair p's rows start
    constexpr int NN_H = APW == 2 ? 1024 : 2048;                                                         // NN: the second k-quad of a region

    floatx4 acc[MT][4]; // [t][u]
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][u][e] = 0.f;
    uintx4 a_r[2][MT]; // [register set][M tile]
    half8_t b_f[2][4];

    // fragment-producing operations of one half-stage (slot pointer sl), into register set `set`.
    // TN: MT reads of A, 4 of B. NN: per pair p 4 transposing reads and, behind them, 4 lane swaps; 4 reads of B -- numbered
    // [pairs 0, 1: 8 reads | B: 4 | pairs 2, 3: 8 reads (TM = 256) | swaps: 4 per pair]
    constexpr int kBReads = B_NC ? 8 : 4; // B_NC: two pairs x (2 tiles x 2 k-quads) transposing reads
    constexpr int kReads = TRANS_A ? MT + 4 : 4 * MP + kBReads;
    constexpr int kOps = TRANS_A || WG_NN_NOSWAP ? kReads : kReads + 4 * MP;
    auto frag_op = [&](const char *sl, int op, int set) {
        auto rb = [&](int u) {
            if constexpr (B_NC) { // read u = 4 p' + 2 tb + h: k-quad h of this lane row's k-group for N tile 2 p' + tb, halves 4 h .. 4 h + 3
                const int pp = u >> 2, tb = (u >> 1) & 1, h = u & 1;
                const uintx2 v = __builtin_bit_cast(uintx2, lds_tr(sl + bn_off[tb] + h * 1024 + pp * 256));
                uintx4 w = __builtin_bit_cast(uintx4, b_f[set][2 * pp + tb]);
                w[2 * h] = v[0]; w[2 * h + 1] = v[1];
                b_f[set][2 * pp + tb] = __builtin_bit_cast(half8_t, w);
            } else b_f[set][u] = lds_h8(sl + b_off + (u >> 1) * RB + (u & 1) * 1024);
        };
        if constexpr (TRANS_A) {
            if (op < MT) a_r[set][op] = __builtin_bit_cast(uintx4, lds_h8(sl + a_off[op & 1] + a_pair_off(op >> 1)));
            else rb(op - MT);
        } else if constexpr (WG_NN_NOSWAP) {
            auto tr = [&](int p, int i) { // read i = 2 tb + h of pair p: the 4 k of k-quad h of this lane row's k-group, tile 2 p + tb, dwords 2 h, 2 h + 1
                const int tb = i >> 1, h = i & 1;
                const uintx2 v = __builtin_bit_cast(uintx2, lds_tr(sl + a_off[tb] + h * NN_H + p * 256));
                a_r[set][2 * p + tb][2 * h] = v[0];
                a_r[set][2 * p + tb][2 * h + 1] = v[1];
            };
            if (op < 8) tr(op >> 2, op & 3);
            else if (op < 8 + kBReads) rb(op - 8);
            else tr(2 + ((op - 8 - kBReads) >> 2), (op - 8 - kBReads) & 3);
        } else {
            auto tr = [&](int p, int i) { // lands in tile 2 p + ins, dwords 2 h, 2 h + 1
                const int h = i >> 1, ins = i & 1;
                const uintx2 v = __builtin_bit_cast(uintx2, lds_tr(sl + a_off[0] + ins * RB + h * NN_H + p * 256));
                a_r[set][2 * p + ins][2 * h] = v[0];
                a_r[set][2 * p + ins][2 * h + 1] = v[1];
            };
            auto sw = [&](int p, int i) { // put the k-groups back on the lane rows the MFMA expects
                const uintx2 r = __builtin_amdgcn_permlane16_swap(a_r[set][2 * p][i], a_r[set][2 * p + 1][i], false, false);
                a_r[set][2 * p][i] = r[0];
                a_r[set][2 * p + 1][i] = r[1];
            };
            if (op < 8) tr(op >> 2, op & 3);
            else if (op < 12) rb(op - 8);
            else if (op < kReads) tr(2 + ((op - 12) >> 2), (op - 12) & 3);
            else sw((op - kReads) >> 2, (op - kReads) & 3);
        }
    };

    const uint32_t NH = K_loc / 32u + (rem ? 2u : 0u); // half-steps (even, >= 2; with a remainder >= 4: the launcher leaves >= 64 whole k)
    // prologue: half-stages 0 .. min(5, NH) - 1; the first two must have landed before the first fragment reads / half-step 0
    if (rem) {
        auto put = [&](uint32_t dst, const char *src, bool valid) {
            uintx4 v = { 0u, 0u, 0u, 0u };
            if (valid) v = *reinterpret_cast<const uintx4_u *>(src);
            *(WG_AS3 uintx4 *)(uintptr_t)(dst + 16u * (uint32_t)lane) = v;
        };
        const char *ra = TRANS_A ? (const char *)(a_base + rem_dk) : (const char *)(a_base + (uint64_t)rem_dk * g.lda);
        const char *rb = (const char *)(b_base + rem_dk);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < APW; ++q) {
                const uint32_t PA = (uint32_t)APW * wave + q, row_a = 16u * PA + (lane >> 2);
                // k-contiguous rows: the lane's 8 k are logical chunk (lane & 3) ^ G(key(row)) of the half-stage's 32
                uint32_t ka;
                if constexpr (TRANS_A) ka = 32u * h + 8u * ((lane & 3u) ^ ((4u - ((row_a >> 3) & 3u)) & 3u));
                else ka = 32u * h + 4u * (APW == 2 ? PA : PA >> 1) + ((lane >> 2) & 3u); // column-major A: the piece's k row
                put(lds_wave + (uint32_t)h * T_SLOT + 1024u * q, ra + (uint64_t)h * a_step + (a_voff[q] - (T_BIAS - 1024u * q)), ka < rem);
                if constexpr (!B_NC) if (q < 2) { // (B_NC: K % 64 == 0 by the launcher: no remainder stage)
                    const uint32_t row_b = 16u * (2u * wave + q) + (lane >> 2);
                    const uint32_t kb = 32u * h + 8u * ((lane & 3u) ^ ((4u - ((row_b >> 2) & 3u)) & 3u));
                    put(lds_wave + (uint32_t)h * T_SLOT + (uint32_t)APW * 1024u + 1024u * q, rb + (uint64_t)h * 64u + (b_voff[q] - (T_BIAS - (uint32_t)B_IMM0 - 1024u * q)), kb < rem);
                }
            }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0); // the loads above are the compiler's to wait for: keep them in front of the DMA pieces it cannot see
#pragma unroll
        for (int h = 2; h < T_RING; ++h)
            if ((uint32_t)h < NH) issue(h, h * T_SLOT);
        // (no DMA piece has to have landed yet: half-step 0 reads slots 0 and 1; its own counted wait covers half-stage 2)
    } else {
#pragma unroll
        for (int h = 0; h < T_RING; ++h)
            if ((uint32_t)h < NH) issue(h, h * T_SLOT);
        t_wait_keep_pieces<T_RING, PPW>(PPW * ((int)min(NH, (uint32_t)T_RING) - 2)); // half-stages 0 and 1 have landed (NH >= 2)
    }
    __syncthreads();
#pragma unroll
    for (int op = 0; op < kOps; ++op) frag_op(smem, op, 0);
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();       // every wave has read slot 0: half-step 0 may refill it
    __builtin_amdgcn_sched_barrier(0);

    uint32_t cur = 0; // byte offset of the slot of half-stage H
    // half-step H on register set SET: 4 MT MFMAs; fragments of H+1 into the other set (NEXT); DMA of H+T_RING into the slot of H (DMA:
    // its fragments were read during H-1 and the barrier that ended H-1 has passed); then wait until at most KEEP of this
    // wave's pieces are in flight (half-stage H+2 has landed; KEEP < 0: nothing to wait for) and publish. The flags are
    // compile-time in the steady state and the peeled tail: as run-time conditions they put a branch around every fragment
    // read (measured: MFMA + reads alone 37 % busy); run-time flags only for K < 192 (fewer than 6 half-steps).
    auto half_step = [&](auto set_c, uint32_t H, auto dma_f, auto next_f, auto keep_f) {
        constexpr int SET = decltype(set_c)::value;
        const uint32_t nxt = cur + T_SLOT == T_RING * T_SLOT ? 0u : cur + T_SLOT;
        const char *sl = smem + nxt;
        if (dma_f()) issue(H + (uint32_t)T_RING, cur);
        t_static_for<4 * MT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int t = j >> 2, u = j & 3;
            if (!(WG_T128_ABLATE & 1))
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a_r[SET][t]), b_f[SET][u], acc[t][u], 0, 0, 0);
            if (next_f() && !(WG_T128_ABLATE & 1)) {
                if constexpr (TRANS_A) {
                    if constexpr (j < kOps) frag_op(sl, j, SET ^ 1);
                } else {
                    constexpr int SW0 = 4 + 2 * MP; // first swap's slot: pair p's swaps sit >= 4 slots behind its reads (TM = 128: 8; 256: 12)
                    if constexpr (j < kReads) frag_op(sl, j, SET ^ 1);                                  // transpose reads and B reads
                    if constexpr (!WG_NN_NOSWAP && j >= SW0 && j < SW0 + 4 * MP) frag_op(sl, kReads + (j - SW0), SET ^ 1); // lane swaps, behind their reads
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): fragments of H+1 are in registers
        constexpr int keep = decltype(keep_f())::value; // pieces of this wave that may stay in flight (half-stage H+2 has landed); < 0: nothing to wait for
        if constexpr (keep > 0) wait_dma_keep<keep>(); else if constexpr (keep == 0) wait_dma_all();
        if (next_f() && !(WG_T128_ABLATE & 4)) __builtin_amdgcn_s_barrier();
        cur = nxt;
    };
    using s0 = std::integral_constant<int, 0>;
    using s1 = std::integral_constant<int, 1>;
    auto yes = [] { return true; };
    auto no = [] { return false; };
    constexpr int kSteady = PPW * (T_RING - 2); // in flight after the steady-state wait: half-stages H+3 .. H+T_RING
    auto keep_c = [](auto n) { return [] { return std::integral_constant<int, decltype(n)::value>{}; }; };
    auto k0 = keep_c(std::integral_constant<int, 0>{});
    auto kn = keep_c(std::integral_constant<int, -1>{});
    if (NH >= (uint32_t)T_RING + 1u) {
        uint32_t H = 0;
        for (; H + (uint32_t)T_RING + 1u < NH; H += 2) {
            half_step(s0{}, H, yes, yes, keep_c(std::integral_constant<int, kSteady>{}));
            half_step(s1{}, H + 1u, yes, yes, keep_c(std::integral_constant<int, kSteady>{}));
        }
        // T_RING + 1 half-steps left (NH is even): H + T_RING = NH - 1 is the last half-stage to issue; step i leaves H+i+3 .. H+T_RING in flight
        t_static_for<T_RING + 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int left = T_RING - 2 - i;
            auto keep = keep_c(std::integral_constant<int, (left > 0 ? PPW * left : left == 0 ? 0 : -1)>{});
            if constexpr (i == 0) half_step(s0{}, H, yes, yes, keep);
            else if constexpr (i == T_RING) half_step(s1{}, H + (uint32_t)i, no, no, keep);
            else if constexpr (i % 2 == 0) half_step(s0{}, H + (uint32_t)i, no, yes, keep);
            else half_step(s1{}, H + (uint32_t)i, no, yes, keep);
        });
    } else {
        for (uint32_t H = 0; H < NH; H += 2) { // NH = 2 .. T_RING - 1: everything was issued by the prologue
            half_step(s0{}, H, no, yes, k0);
            half_step(s1{}, H + 1u, no, [&] { return H + 2u < NH; }, kn);
        }
    }

    // ---- epilogue: lane holds, per (pair p, N tile u), rows 32 p + 8 kg + 0..7 of column 16 u + i16 of the wave tile (B_NC: of column 32 (u >> 1) + 8 a + 4 ((u & 1) ^ (a & 1)) + e, i16 = 4 a + e) ----
    auto col_in_wave = [&](int u) -> uint32_t {
        if constexpr (B_NC) { const uint32_t ca = (uint32_t)i16 >> 2; return 32u * (uint32_t)(u >> 1) + 8u * ca + 4u * ((uint32_t)(u & 1) ^ (ca & 1u)) + ((uint32_t)i16 & 3u); }
        else return 16u * (uint32_t)u + (uint32_t)i16;
    };
    const bool full_tile = (m0 + TM <= g.M) && (n0 + TN <= g.N);
    const uint32_t row0 = m0 + (uint32_t)(TM / 2) * wm + 8u * kg;
    if (g.nsplit > 1) { // split-K: raw f32 partial sums to this split's slab (dense, ld = M); wg_splitk_reduce finishes
        float *P = g.part + ((uint64_t)z * g.nsplit + split) * ((uint64_t)g.M * g.N);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t col = n0 + 64u * wn + col_in_wave(u);
            if (!full_tile && col >= g.N) continue;
            float *pc = P + (uint64_t)col * g.M + row0;
#pragma unroll
            for (int p = 0; p < MP; ++p) {
                if (!(full_tile || row0 + 32 * p < g.M)) continue;
                float4 *d = reinterpret_cast<float4 *>(pc + 32 * p);
                d[odd_row ? 1 : 0] = make_float4(acc[2 * p][u][0], acc[2 * p][u][1], acc[2 * p][u][2], acc[2 * p][u][3]);
                d[odd_row ? 0 : 1] = make_float4(acc[2 * p + 1][u][0], acc[2 * p + 1][u][1], acc[2 * p + 1][u][2], acc[2 * p + 1][u][3]);
            }
        }
        return;
    }
    _Float16 *C = g.c + z * g.c_batch;
    const float alpha = g.alpha, beta = g.beta;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t col = n0 + 64u * wn + col_in_wave(u);
        if (!full_tile && col >= g.N) continue;
        _Float16 *cc = C + (uint64_t)col * g.ldc + row0;
#pragma unroll
        for (int p = 0; p < MP; ++p) {
            if (!(full_tile || row0 + 32 * p < g.M)) continue; // 8 consecutive rows, all in or all out (M % 8 == 0)
            float r[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { // (odd lane rows of the swap-free Gemm: the pair's tiles in exchanged order)
                r[q] = odd_row ? acc[2 * p + 1][u][q] : acc[2 * p][u][q];
                r[4 + q] = odd_row ? acc[2 * p][u][q] : acc[2 * p + 1][u][q];
            }
            if (alpha != 1.f) {
#pragma unroll
                for (int q = 0; q < 8; ++q) r[q] *= alpha;
            }
            if (beta != 0.f) { // beta == 0 never reads C
                const half8_t c = *reinterpret_cast<const half8_u *>(cc + 32 * p);
#pragma unroll
                for (int q = 0; q < 8; ++q) r[q] = fmaf(beta, (float)c[q], r[q]);
            }
            half8_t v;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (_Float16)r[q];
            *reinterpret_cast<half8_u *>(cc + 32 * p) = v;
        }
    }
}

} // namespace

// the B_NC instances (row-major GemmTr on mid-size outputs: gemm_f16_nt.hip's launcher). Whole K per workgroup, K % 64 == 0.
int t128_launch_nt(wg_ctx *ctx, dim3 grid, const GemmArgs &g, int tm) {
    if (tm == 256) hipLaunchKernelGGL((gemm_f16_t128_kernel<false, 256, true>), grid, dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL((gemm_f16_t128_kernel<false, 128, true>), grid, dim3(256), 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

int t128_launch(wg_ctx *ctx, bool trans, dim3 grid, const GemmArgs &g, int tm) {
    if (tm == 256) {
        if (trans) hipLaunchKernelGGL((gemm_f16_t128_kernel<true, 256>), grid, dim3(256), 0, ctx->stream, g);
        else hipLaunchKernelGGL((gemm_f16_t128_kernel<false, 256>), grid, dim3(256), 0, ctx->stream, g);
    } else if (trans) hipLaunchKernelGGL((gemm_f16_t128_kernel<true, 128>), grid, dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL((gemm_f16_t128_kernel<false, 128>), grid, dim3(256), 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

} // namespace wgf16
