// f16 Gemm with BOTH operands contiguous along their output dimension ("NT"): out (M x N, column-major) = A (M x K, m-contiguous) * B (K x N, n-contiguous:
// element (k, n) at b + n + k * ldb). This is what the ROW-MAJOR GemmTr of the reference is in column-major terms (shape.wgsl:49-57 with
// row_major_shader_defs(), shape.rs:13-15; gemm.wgsl:115-148): out = m1^T m2 on row-major views <=> out^T (N x M) = m2^T (N x K, "m"-contiguous) * m1 (K x M,
// "n"-contiguous). Round 5 transposed m1 into a scratch buffer first (one extra HBM-bound pass, api.hip); this kernel takes the operands where they lie.
//
// Same machine as gemm_f16.hip's 16x16x32 kernel -- 256 x 256 block tile, 2 x 2 waves of 128 x 128 = 8 x 8 MFMA tiles (v_mfma_f32_16x16x32_f16), f32 accumulation,
// one RNE rounding, operands by LDS-DMA (global_load_lds_dwordx4), fragments of half-step H + 1 read during H -- with B on the path Gemm's column-major A takes there:
//   * LDS: A ring 4 x 16 KiB + B ring 4 x 16 KiB of HALF-stages (32 k): 128 KiB. Both rings turn together (one read offset, one DMA offset).
//   * layout of both operands: 256-byte blocks [k/4][x/32][4 k][32 x] (x = m or n) read with ds_read_b64_tr_b16, swap-free: the tiles of a pair interleave by
//     unit parity and the pieces of odd k-groups swap neighbouring 16-byte units (gemm_f16.hip, "NN A"). Lane (kg, i16) then holds, for M tile 2 p + tb, rows
//     32 p + 8 kg + 4 (tb ^ (kg & 1)) + r, and, for N tile 2 p' + tb, COLUMN 32 p' + 8 a + 4 (tb ^ (a & 1)) + e (i16 = 4 a + e): the epilogue's column index.
//   * per half-step and wave: 64 MFMAs, 32 transposing reads (16 A + 16 B), 8 DMA pieces (4 A, then 4 B, of half-stage H + 4 into the slot H released), one
//     lgkmcnt(0) + vmcnt(16) + barrier. The k order inside every MFMA and the accumulation chain of every element are those of the column-major kernels: the result
//     is bit-identical to transposing B first and calling them (tests/test_gpu_parity.py).
// Takes: M % 8 == 0, N % 8 == 0, K % 64 == 0, K >= 256 (any leading dimension and offset since round 6); everything else goes the transposed-copy way (api.hip).
// Outputs of fewer 256 x 256 tiles than CUs are handed to the 128 x 128 / 256 x 128 tiles of gemm_f16_t128.hip in their B_NC instances (the launcher below).
// Bound: MFMA (2.5 PFLOP/s dense), in practice the package power cap, as the other f16 kernels.
#include "gemm_f16_common.hpp"

namespace wgf16 {
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void nt_static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void nt_static_for(F &&f) { nt_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr uint32_t NT_BIAS = 3072;         // see M16_BIAS in gemm_f16.hip: scalar bases lowered by this, voff of piece q raised by BIAS - 1024 q
constexpr uint32_t NT_B_BASE = 4u * HA_BYTES; // 64 KiB: the B ring behind the A ring
constexpr int NT_DO = 3, NT_DS = 8, NT_SYNC = 60, NT_KEEP = 16;

__device__ __forceinline__ void nt_set_m0(uint32_t lds_dst) { asm volatile("s_mov_b32 m0, %0" ::"s"(lds_dst)); }
template <int IMM>
__device__ __forceinline__ void nt_dma(uint32_t voff, const void *sbase) {
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%c2" ::"v"(voff), "s"(sbase), "i"(IMM));
}

__global__ __launch_bounds__(256, 1) void gemm_f16_nt_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i16 = lane & 15, kg = lane >> 4;
    uint32_t tm, tn;
    tile_of(blockIdx.x, g.tiles_m, g.tiles_n, tm, tn);
    tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
    const uint32_t m0 = tm * BM, n0 = tn * BN, z = blockIdx.y;
    auto sc64 = [](uint64_t v) -> uint64_t { // (64-bit products run on the vector unit even when uniform: back to scalar registers by hand -- they end up in scalar operands of the DMA asm)
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const _Float16 *A = g.a + sc64((uint64_t)z * g.a_batch) + m0;
    const _Float16 *B = g.b + sc64((uint64_t)z * g.b_batch) + n0;
    _Float16 *C = g.c + sc64((uint64_t)z * g.c_batch);
    float alpha = g.alpha, beta = g.beta;
    uint32_t ldc = g.ldc;
    asm volatile("" : "+s"(C), "+s"(alpha), "+s"(beta), "+s"(ldc)); // (pinned now: scalar loads in front of the loop would share lgkmcnt with its LDS reads)

    // ---- DMA addressing: a half-stage of either operand = 16 pieces of 1 KiB, wave stages P = 4 wave + q: k-quad kq = P >> 1 (its k-group within the half-stage is
    // `wave`), x-half P & 1; lane -> block 4 (P & 1) + (lane >> 4), k row (lane >> 2) & 3, 16-byte unit lane & 3 -- fetched from unit (lane & 3) ^ (wave & 1). Rows /
    // columns past the end of a ragged tile are clamped to the last valid piece (results never stored).
    uint32_t a_voff[4], b_voff[4];
    const uint32_t unit = (lane & 3u) ^ ((uint32_t)wave & 1u);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t P = 4u * wave + q;
        const uint32_t x = 128u * (P & 1) + 32u * (lane >> 4) + 8u * unit, krow = 4u * (P >> 1) + ((lane >> 2) & 3);
        a_voff[q] = (krow * g.lda + min(x, g.M - 8u - m0)) * 2u + (NT_BIAS - 1024u * q);
        b_voff[q] = (krow * g.ldb + min(x, g.N - 8u - n0)) * 2u + (NT_BIAS - 1024u * q);
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(WG_AS3 char *)smem;
    const uint32_t lds_a_wave = __builtin_amdgcn_readfirstlane(lds_base + wave * 4096);
    const uint32_t lds_b_wave = __builtin_amdgcn_readfirstlane(lds_base + NT_B_BASE + wave * 4096);
    const uint32_t a_step = BKH * g.lda * 2u, b_step = BKH * g.ldb * 2u; // bytes between two half-stages (< 2^32: launcher)
    // ---- per-lane LDS read addresses: lane row kg reads k-group kg (blocks kq = 2 kg + h); lane 4 krow + a: k row krow, unit a -- at position a ^ (kg & 1) --, of it
    // the 8-byte half tb ^ (a & 1) for tile tb of a pair (base 0: even tiles, base 1: odd tiles); + h * 2048 + pair * 256 per read
    const uint32_t ua = (uint32_t)i16 & 3u, krow_l = (uint32_t)i16 >> 2;
    const uint32_t common = (uint32_t)kg * 4096u + krow_l * 64u + (ua ^ ((uint32_t)kg & 1u)) * 16u;
    const uint32_t vA0 = lds_base + common + (4u * wm) * 256u + (ua & 1u) * 8u, vA1 = lds_base + common + (4u * wm) * 256u + ((ua & 1u) ^ 1u) * 8u;
    const uint32_t vB0 = lds_base + NT_B_BASE + common + (4u * wn) * 256u + (ua & 1u) * 8u, vB1 = lds_base + NT_B_BASE + common + (4u * wn) * 256u + ((ua & 1u) ^ 1u) * 8u;

    floatx4 acc[8][8]; // [M tile t][N tile u]
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][u][e] = 0.f;
    uintx4 a_r[2][8], b_r[2][8]; // [register set][tile]: bit patterns of 8 halves

    const uint32_t NH = g.K / 32u; // half-stages (>= 8)
    uint32_t va = vA0, va1 = vA1, vb = vB0, vb1 = vB1; // where this half-step's fragment reads start
    uint32_t rR = 1u << 14, rD = 0;                   // ring offsets (4 slots of 16 KiB): read by this half-step / DMA'd by this half-step
    uint32_t lb = 0;
    const char *ga, *gb; // SGPR pairs: global bases (less NT_BIAS) of the pieces issued this half-step
    // cursor steps: 0 once the next piece would lie past the last half-stage (the cursors then stay parked on it: its pieces land in slots nobody reads any more, and
    // every counted wait means the same all the way -- gemm_f16.hip's unpeeled loop)
    uint32_t hn = 0, a_inc = a_step, b_inc = b_step; // hn: the half-step that starts next; a cursor step after half-step H is taken iff H + 5 < NH (NH >= 8)
    uint32_t st = 0;
    const uint32_t S = NH / 2u;

    auto frag = [&](int op, int set) {
        auto tr = [&](uintx4 &dst, uint32_t base, int p, int h) {
            const uintx2 v = __builtin_bit_cast(uintx2, lds_tr_at(base + h * 2048 + p * 256));
            dst[2 * h] = v[0];
            dst[2 * h + 1] = v[1];
        };
        auto ta = [&](int p, int i) { tr(a_r[set][2 * p + (i >> 1)], (i >> 1) ? va1 : va, p, i & 1); };
        auto tb = [&](int p, int i) { tr(b_r[set][2 * p + (i >> 1)], (i >> 1) ? vb1 : vb, p, i & 1); };
        // in the order the next half-step's MFMAs (t = j >> 3, u = j & 7) want them: A 0, 1 | B 0..3 | A 2, 3 | B 4..7 | A 4..7
        if (op < 4) ta(0, op);
        else if (op < 12) tb((op - 4) >> 2, (op - 4) & 3);
        else if (op < 16) ta(1, op - 12);
        else if (op < 24) tb(2 + ((op - 16) >> 2), (op - 16) & 3);
        else ta(2 + ((op - 24) >> 2), (op - 24) & 3);
    };
    constexpr int kOps = 32;

    auto half_step = [&](auto hs_c) {
        constexpr int HS = decltype(hs_c)::value;
        nt_static_for<64>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int t = j >> 3, u = j & 7;
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a_r[HS][t]), __builtin_bit_cast(half8_t, b_r[HS][u]), acc[t][u], 0, 0, 0);
            if constexpr ((j & 3) != 3 && 3 * (j >> 2) + (j & 3) < kOps) frag(3 * (j >> 2) + (j & 3), HS ^ 1); // three fragment ops in every four slots (0 .. 41)
            if constexpr (j == NT_DO - 1) nt_set_m0(lds_a_wave + rD);
            if constexpr (j == NT_DO + 4 * NT_DS - 5) { lb = lds_b_wave + rD; asm volatile("" : "+s"(lb)); }
            if constexpr (j == NT_DO + 4 * NT_DS - 1) nt_set_m0(lb);
            if constexpr (j >= NT_DO && ((j - NT_DO) % NT_DS) == 0 && (j - NT_DO) / NT_DS < 8) {
                constexpr int pi = (j - NT_DO) / NT_DS, q = pi & 3;
                if constexpr (pi < 4) nt_dma<1024 * q>(a_voff[q], ga);
                else nt_dma<1024 * q>(b_voff[q], gb);
            }
            if constexpr (j == NT_DO + 3 * NT_DS + 1) { ga += a_inc; asm volatile("" : "+s"(ga)); }                   // after the last A piece
            if constexpr (j == NT_DO + 4 * NT_DS + 1) { rD = (rD + 0x4000u) & 0xffffu; asm volatile("" : "+s"(rD)); } // after B's M0
            if constexpr (j == 47) { rR = (rR + 0x4000u) & 0xffffu; asm volatile("" : "+s"(rR)); }                    // after the last fragment read (slot 41)
            if constexpr (j == 49) { va = vA0 + rR; asm volatile("" : "+v"(va)); }
            if constexpr (j == 50) { va1 = vA1 + rR; asm volatile("" : "+v"(va1)); }
            if constexpr (j == 52) { vb = vB0 + rR; asm volatile("" : "+v"(vb)); }
            if constexpr (j == 53) { vb1 = vB1 + rR; asm volatile("" : "+v"(vb1)); }
            if constexpr (j == 54) { ++hn; asm volatile("" : "+s"(hn)); }
            if constexpr (j == 55) { a_inc = hn + 5u < NH ? a_step : 0u; asm volatile("" : "+s"(a_inc)); }
            if constexpr (j == 56 && HS == 1) { ++st; asm volatile("" : "+s"(st)); }
            if constexpr (j == NT_SYNC) { // every fragment read and every piece of the half-step is issued: lgkmcnt(0), the pieces of the last two half-steps may fly, publish
                __builtin_amdgcn_s_waitcnt(0xc07f);
                wait_dma_keep<NT_KEEP>();
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (j == NT_SYNC + 1) { gb += b_inc; asm volatile("" : "+s"(gb)); } // after the last B piece (slot 59)
            if constexpr (j == NT_SYNC + 2) { b_inc = hn + 5u < NH ? b_step : 0u; asm volatile("" : "+s"(b_inc)); }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- prologue: half-stages 0 .. 3 of both operands, in the order the counted waits expect them to retire ----
    {
        const char *pa = (const char *)A - NT_BIAS, *pb = (const char *)B - NT_BIAS; // (cursors advanced by 32-bit steps: a 64-bit product would be computed on the vector unit)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            nt_set_m0(lds_a_wave + h * HA_BYTES); asm volatile("s_nop 0");
            nt_dma<0>(a_voff[0], pa); nt_dma<1024>(a_voff[1], pa); nt_dma<2048>(a_voff[2], pa); nt_dma<3072>(a_voff[3], pa);
            nt_set_m0(lds_b_wave + h * HA_BYTES); asm volatile("s_nop 0");
            nt_dma<0>(b_voff[0], pb); nt_dma<1024>(b_voff[1], pb); nt_dma<2048>(b_voff[2], pb); nt_dma<3072>(b_voff[3], pb);
            pa += a_step; pb += b_step;
        }
        ga = pa; gb = pb; // half-stage 4
    }
    wait_dma_keep<16>(); // half-stages 0 and 1 have landed
    __syncthreads();
#pragma unroll
    for (int op = 0; op < kOps; ++op) frag(op, 0); // half-step 0's own fragments (slot 0)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier(); // every wave has read slot 0: half-step 0 may refill it
    va = vA0 + rR; va1 = vA1 + rR; vb = vB0 + rR; vb1 = vB1 + rR; // half-step 0 reads half-stage 1
    asm volatile("" : "+s"(a_inc), "+s"(b_inc)); // (half-step 0: 5 < NH)
    __builtin_amdgcn_sched_barrier(0);

    while (st < S) { // (a `while` over straight-line pairs of half-steps: see gemm_f16.hip on why not a do-while and why no parity branches)
        half_step(std::integral_constant<int, 0>{});
        half_step(std::integral_constant<int, 1>{}); // ++st inside
    }

    // ---- epilogue: lane holds, per (pair p, N tile u), rows 32 p + 8 kg + 0..7 (odd lane rows: the pair's tiles in exchanged order) of ONE column ----
    const bool full_tile = (m0 + BM <= g.M) && (n0 + BN <= g.N);
    const bool odd_row = (kg & 1) != 0;
    const uint32_t row0 = m0 + 128u * wm + 8u * kg;
    const uint32_t ca = (uint32_t)i16 >> 2; // this lane's MFMA column i16 = 4 ca + e: N tile 2 p' + tb holds column 32 p' + 8 ca + 4 (tb ^ (ca & 1)) + e there
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const uint32_t col = n0 + 128u * wn + 32u * (uint32_t)(u >> 1) + 8u * ca + 4u * ((uint32_t)(u & 1) ^ (ca & 1u)) + ((uint32_t)i16 & 3u);
        if (!full_tile && col >= g.N) continue;
        _Float16 *cc = C + (uint64_t)col * ldc + row0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!(full_tile || row0 + 32 * p < g.M)) continue; // 8 consecutive rows, all in or all out (M % 8 == 0)
            float r[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // nothing (parked pieces, stores) is in flight when the workgroup ends
}

} // namespace

// WG_ERR_UNSUPPORTED (no message): not a product this kernel takes -- the caller goes the transposed-copy way.
static int nt_launch(wg_ctx *ctx, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats, _Float16 *out, uint32_t out_ld, uint64_t out_batch, const _Float16 *a, uint32_t lda,
              uint64_t a_batch, const _Float16 *b, uint32_t ldb, uint64_t b_batch, float alpha, float beta) {
    if (M == 0 || N == 0 || nmats == 0) return WG_OK;
    // (any leading dimension, offset and batch stride: the LDS-DMA and the 16-byte stores take element-aligned addresses -- gemm_f16_common.hpp half8_u)
    if (M % 8u || N % 8u || K % 64u || K < 256u || nmats > 65535u) return WG_ERR_UNSUPPORTED;
    if ((uint64_t)lda * 64u >= (1ull << 31) || (uint64_t)ldb * 64u >= (1ull << 31)) return WG_ERR_UNSUPPORTED; // 32-bit offsets within a half-stage, 32-bit cursor steps
    GemmArgs g{};
    g.a = a; g.lda = lda; g.a_batch = a_batch;
    g.b = b; g.ldb = ldb; g.b_batch = b_batch;
    g.c = out; g.ldc = out_ld; g.c_batch = out_batch;
    g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta;
    g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
    g.nsplit = 1; g.k_per_split = K;
    const uint64_t tiles = (uint64_t)g.tiles_m * g.tiles_n;
    if (tiles > 0x7fffffffull) return WG_ERR_UNSUPPORTED;
    // Fewer 256 x 256 tiles than CUs: the mid-size tiles of gemm_f16_t128.hip in their n-contiguous-B form (two workgroups per CU) -- 128 x 128, or 256 x 128 where
    // that tile alone on its CU measured ahead. WG_TUNE_F16_TILE forces a family.
    const uint64_t cus = (uint64_t)(ctx->compute_units > 0 ? ctx->compute_units : 256);
    const int forced = ctx->tuning[WG_TUNE_F16_TILE];
    if ((forced == 0 && tiles * nmats < cus) || forced == 128 || forced == 256128) {
        const uint64_t t256x128 = (uint64_t)((M + 255u) / 256u) * ((N + 127u) / 128u) * nmats;
        // (the column-major launcher's measured rule for that tile: about one 256 x 128 tile per CU -- 70 .. 100 % of them --, K = 512 .. 4096, one or two matrices)
        const bool tall = forced == 256128 || (forced == 0 && t256x128 <= cus && 10u * t256x128 >= 7u * cus && K >= 512u && K <= 4096u && nmats <= 2u);
        GemmArgs t = g;
        t.tiles_m = tall ? (M + 255u) / 256u : (M + 127u) / 128u;
        t.tiles_n = (N + 127u) / 128u;
        t.part = nullptr;
        const uint64_t tt = (uint64_t)t.tiles_m * t.tiles_n;
        if (tt > 0x7fffffffull) return WG_ERR_UNSUPPORTED;
        return t128_launch_nt(ctx, dim3((uint32_t)tt, nmats), t, tall ? 256 : 128);
    }
    hipLaunchKernelGGL(gemm_f16_nt_kernel, dim3((uint32_t)tiles, nmats), dim3(256), 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

} // namespace wgf16

int wgk_gemm_f16_nt(wg_ctx *ctx, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats, __half *out, uint32_t out_ld, uint64_t out_batch, wgk_mat a_mcontig, wgk_mat b_ncontig,
                    float alpha, float beta) {
    return wgf16::nt_launch(ctx, M, N, K, nmats, (_Float16 *)out, out_ld, out_batch, (const _Float16 *)a_mcontig.ptr, a_mcontig.ld, a_mcontig.batch,
                            (const _Float16 *)b_ncontig.ptr, b_ncontig.ld, b_ncontig.batch, alpha, beta);
}
