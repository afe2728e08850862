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
//   * tile order: 4-tile-row strips, contiguous ranges per XCD (tile_strips, gemm_f16_common.hpp), so an XCD's L2 sees a compact
//     patch of the output.
// M0 is written by the inline asm without save/restore, as in gemm_f16.hip (tests/test_abi_and_host.py checks the ISA).
// Bound: LDS bandwidth (per CU and half-step pair: 64 KiB of fragment reads + 32 KiB of DMA writes against 512 MFMA cycles per
// SIMD) and L2 -> CU bandwidth; see DESIGN.md section 3.
#include "gemm_f16_common.hpp"

namespace wgf16 {
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void t_static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void t_static_for(F &&f) { t_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

#ifndef WG_T128_ABLATE
#define WG_T128_ABLATE 0 // timing experiments only (results are garbage): 1 = no MFMA / fragment reads, 2 = no DMA, 4 = no barriers
#endif
constexpr int TM = 128, TN = 128;
constexpr int T_RING = 5;
constexpr int T_SLOT = 16 * 1024;   // one half-stage: per wave w a 4 KiB region [A piece 2w | A piece 2w+1 | B piece 2w | B piece 2w+1]
constexpr uint32_t T_BIAS = 3072;   // see M16_BIAS in gemm_f16.hip

__device__ __forceinline__ void t_set_m0(uint32_t lds_dst) { asm volatile("s_mov_b32 m0, %0" ::"s"(lds_dst)); }
template <int IMM>
__device__ __forceinline__ void t_dma(uint32_t voff, const void *sbase) {
    if (WG_T128_ABLATE & 2) return;
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%c2" ::"v"(voff), "s"(sbase), "i"(IMM));
}

template <bool TRANS_A>
__global__ __launch_bounds__(256, 2) void gemm_f16_t128_kernel(GemmArgs g) {
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
    const _Float16 *B = g.b + z * g.b_batch + k_begin;

    // ---- DMA addressing: per half-stage this wave stages pieces P = 2 wave + q (q = 0, 1) of A and of B; a piece = 1 KiB of LDS
    // (64 lanes x 16 bytes). k-contiguous operands (B; op(A) for TN): piece P = rows 16P..16P+15 of 64 bytes (32 k), lane -> row
    // 16P + (lane>>2), position lane&3 holds the logical chunk (lane&3) ^ G(key(row)), key = (row>>2)&3 for B (rows read in natural
    // order) and (row>>3)&3 for op(A) (rows read permuted). Column-major A: piece P = k-quad kq = P, blocks [mblk = lane>>4] of
    // [4 k][32 m]: k row (lane>>2)&3, 16-byte unit lane&3. Rows past the end of a ragged tile are clamped (results never stored).
    uint32_t a_voff[2], b_voff[2];
    const _Float16 *a_base, *b_base = B + (uint64_t)n0 * g.ldb;
    if constexpr (TRANS_A) a_base = A + (uint64_t)m0 * g.lda; else a_base = A + m0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t P = 2u * wave + q;
        const uint32_t row = 16u * P + (lane >> 2);
        if constexpr (TRANS_A) {
            const uint32_t ra = min(row, g.M - 1u - m0);
            a_voff[q] = (ra * g.lda + 8u * ((lane & 3u) ^ ((4u - ((row >> 3) & 3u)) & 3u))) * 2u + (T_BIAS - 1024u * q);
        } else {
            const uint32_t k = 4u * P + ((lane >> 2) & 3u);
            const uint32_t m = min(32u * (lane >> 4) + 8u * (lane & 3u), g.M - 8u - m0); // M % 8 == 0
            a_voff[q] = (k * g.lda + m) * 2u + (T_BIAS - 1024u * q);
        }
        const uint32_t rb = min(row, g.N - 1u - n0);
        b_voff[q] = (rb * g.ldb + 8u * ((lane & 3u) ^ ((4u - ((row >> 2) & 3u)) & 3u))) * 2u + (T_BIAS - 1024u * (2 + q));
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(WG_AS3 char *)smem;
    const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(lds_base + wave * 4096);
    const uint64_t a_step = TRANS_A ? 64u : (uint64_t)64u * g.lda; // bytes per half-stage (32 k)
    const char *ga0 = (const char *)a_base - T_BIAS - (rem ? 2u * a_step : 0u), *gb0 = (const char *)b_base - T_BIAS - (rem ? 128u : 0u);
    auto issue = [&](uint32_t H, uint32_t slot_off) { // the 4 pieces of half-stage H into the slot at byte offset slot_off
        const char *ga = ga0 + H * a_step, *gb = gb0 + (uint64_t)H * 64u;
        t_set_m0(lds_wave + slot_off);
        asm volatile("s_nop 0");
        t_dma<0>(a_voff[0], ga); t_dma<1024>(a_voff[1], ga); t_dma<2048>(b_voff[0], gb); t_dma<3072>(b_voff[1], gb);
    };

    // ---- per-lane LDS read offsets within a slot ----
    // B tile u: row 64 wn + 16 u + i16 = piece 4 wn + u, row i16 of it -> (2 wn + (u>>1)) * 4096 + 2048 + (u&1) * 1024 + i16 * 64 + pos * 16
    const uint32_t pos = (uint32_t)(kg ^ gq);
    const uint32_t b_off = 2u * wn * 4096u + 2048u + (uint32_t)i16 * 64u + pos * 16u;
    // A, TN: MFMA tile t = 2 p + tb, MFMA row i16 <-> tile row 64 wm + 32 p + 8 aq + 4 tb + bb = piece 4 wm + 2 p + (aq>>1), row 8 (aq&1) + 4 tb + bb
    // A, NN: transpose read i = 2 h + ins of pair p: k-quad kq = 4 (kg>>1) + 2 ins + h (piece kq), block mblk = 2 wm + p, unit i16, half kg&1
    uint32_t a_off[2];
    if constexpr (TRANS_A) {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
            a_off[tb] = 2u * wm * 4096u + (uint32_t)(aq >> 1) * 1024u + (uint32_t)(8 * (aq & 1) + 4 * tb + bb) * 64u + pos * 16u;
    } else {
        a_off[0] = (uint32_t)(2 * (kg >> 1)) * 4096u + (2u * wm) * 256u + (uint32_t)i16 * 16u + (uint32_t)(kg & 1) * 8u;
        a_off[1] = 0;
    }

    floatx4 acc[4][4]; // [t][u]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][u][e] = 0.f;
    uintx4 a_r[2][4]; // [register set][M tile]
    half8_t b_f[2][4];

    // fragment-producing operations of one half-stage (slot pointer sl), into register set `set`
    constexpr int kOps = TRANS_A ? 8 : 20;
    auto frag_op = [&](const char *sl, int op, int set) {
        auto rb = [&](int u) { b_f[set][u] = lds_h8(sl + b_off + (u >> 1) * 4096 + (u & 1) * 1024); };
        if constexpr (TRANS_A) {
            if (op < 4) a_r[set][op] = __builtin_bit_cast(uintx4, lds_h8(sl + a_off[op & 1] + (op >> 1) * 4096));
            else rb(op - 4);
        } else {
            auto tr = [&](int p, int i) { // lands in tile 2 p + ins, dwords 2 h, 2 h + 1
                const int h = i >> 1, ins = i & 1;
                const uintx2 v = __builtin_bit_cast(uintx2, lds_tr(sl + a_off[0] + ins * 4096 + h * 1024 + p * 256));
                a_r[set][2 * p + ins][2 * h] = v[0];
                a_r[set][2 * p + ins][2 * h + 1] = v[1];
            };
            auto sw = [&](int p, int i) { // put the k-groups back on the lane rows the MFMA expects
                const uintx2 r = __builtin_amdgcn_permlane16_swap(a_r[set][2 * p][i], a_r[set][2 * p + 1][i], false, false);
                a_r[set][2 * p][i] = r[0];
                a_r[set][2 * p + 1][i] = r[1];
            };
            if (op < 4) tr(0, op);
            else if (op < 8) tr(1, op - 4);
            else if (op < 12) rb(op - 8);
            else if (op < 16) sw(0, op - 12);
            else sw(1, op - 16);
        }
    };

    const uint32_t NH = K_loc / 32u + (rem ? 2u : 0u); // half-steps (even, >= 2; with a remainder >= 4: the launcher leaves >= 64 whole k)
    // prologue: half-stages 0 .. min(5, NH) - 1; the first two must have landed before the first fragment reads / half-step 0
    if (rem) {
        auto put = [&](uint32_t dst, const char *src, bool valid) {
            uintx4 v = { 0u, 0u, 0u, 0u };
            if (valid) v = *reinterpret_cast<const uintx4 *>(src);
            *(WG_AS3 uintx4 *)(uintptr_t)(dst + 16u * (uint32_t)lane) = v;
        };
        const char *ra = TRANS_A ? (const char *)(a_base + rem_dk) : (const char *)(a_base + (uint64_t)rem_dk * g.lda);
        const char *rb = (const char *)(b_base + rem_dk);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint32_t P = 2u * wave + q, row = 16u * P + (lane >> 2);
                // k-contiguous rows: the lane's 8 k are logical chunk (lane & 3) ^ G(key(row)) of the half-stage's 32
                const uint32_t kb = 32u * h + 8u * ((lane & 3u) ^ ((4u - ((row >> 2) & 3u)) & 3u));
                uint32_t ka;
                if constexpr (TRANS_A) ka = 32u * h + 8u * ((lane & 3u) ^ ((4u - ((row >> 3) & 3u)) & 3u));
                else ka = 32u * h + 4u * P + ((lane >> 2) & 3u); // column-major A: the piece's k row
                put(lds_wave + (uint32_t)h * T_SLOT + 1024u * q, ra + (uint64_t)h * a_step + (a_voff[q] - (T_BIAS - 1024u * q)), ka < rem);
                put(lds_wave + (uint32_t)h * T_SLOT + 2048u + 1024u * q, rb + (uint64_t)h * 64u + (b_voff[q] - (T_BIAS - 1024u * (2 + q))), kb < rem);
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
        if (NH >= 5) wait_dma_keep<12>(); else if (NH == 4) wait_dma_keep<8>(); else wait_dma_all();
    }
    __syncthreads();
#pragma unroll
    for (int op = 0; op < kOps; ++op) frag_op(smem, op, 0);
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();       // every wave has read slot 0: half-step 0 may refill it
    __builtin_amdgcn_sched_barrier(0);

    uint32_t cur = 0; // byte offset of the slot of half-stage H
    // half-step H on register set SET: 16 MFMAs; fragments of H+1 into the other set (NEXT); DMA of H+5 into the slot of H (DMA:
    // its fragments were read during H-1 and the barrier that ended H-1 has passed); then wait until at most KEEP of this
    // wave's pieces are in flight (half-stage H+2 has landed; KEEP < 0: nothing to wait for) and publish. The flags are
    // compile-time in the steady state and the peeled tail: as run-time conditions they put a branch around every fragment
    // read (measured: MFMA + reads alone 37 % busy); run-time flags only for K < 192 (fewer than 6 half-steps).
    auto half_step = [&](auto set_c, uint32_t H, auto dma_f, auto next_f, auto keep_f) {
        constexpr int SET = decltype(set_c)::value;
        const uint32_t nxt = cur + T_SLOT == T_RING * T_SLOT ? 0u : cur + T_SLOT;
        const char *sl = smem + nxt;
        if (dma_f()) issue(H + 5u, cur);
        t_static_for<16>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int t = j >> 2, u = j & 3;
            if (!(WG_T128_ABLATE & 1))
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a_r[SET][t]), b_f[SET][u], acc[t][u], 0, 0, 0);
            if (next_f() && !(WG_T128_ABLATE & 1)) {
                if constexpr (TRANS_A) {
                    if constexpr (j < kOps) frag_op(sl, j, SET ^ 1);
                } else {
                    if constexpr (j < 12) frag_op(sl, j, SET ^ 1);               // 8 transpose reads, 4 B reads
                    if constexpr (j >= 8 && j < 16) frag_op(sl, j + 4, SET ^ 1);  // 8 lane swaps, behind their reads
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): fragments of H+1 are in registers
        const int keep = keep_f();
        if (keep == 12) wait_dma_keep<12>(); else if (keep == 8) wait_dma_keep<8>(); else if (keep == 4) wait_dma_keep<4>(); else if (keep == 0) wait_dma_all();
        if (next_f() && !(WG_T128_ABLATE & 4)) __builtin_amdgcn_s_barrier();
        cur = nxt;
    };
    using s0 = std::integral_constant<int, 0>;
    using s1 = std::integral_constant<int, 1>;
    auto yes = [] { return true; };
    auto no = [] { return false; };
    auto k12 = [] { return 12; };
    auto k8 = [] { return 8; };
    auto k4 = [] { return 4; };
    auto k0 = [] { return 0; };
    auto kn = [] { return -1; };
    if (NH >= 6) {
        uint32_t H = 0;
        for (; H + 6u < NH; H += 2) {
            half_step(s0{}, H, yes, yes, k12);
            half_step(s1{}, H + 1u, yes, yes, k12);
        }
        // six half-steps left (NH is even): H+5 = NH-1 is the last half-stage to issue
        half_step(s0{}, H, yes, yes, k12);      // in flight after the wait: H+3, H+4, H+5
        half_step(s1{}, H + 1u, no, yes, k8);   // H+4, H+5
        half_step(s0{}, H + 2u, no, yes, k4);   // H+5
        half_step(s1{}, H + 3u, no, yes, k0);
        half_step(s0{}, H + 4u, no, yes, kn);
        half_step(s1{}, H + 5u, no, no, kn);
    } else {
        for (uint32_t H = 0; H < NH; H += 2) { // NH = 2 or 4: everything was issued by the prologue
            half_step(s0{}, H, no, yes, k0);
            half_step(s1{}, H + 1u, no, [&] { return H + 2u < NH; }, kn);
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
