// f32 Gemm / GemmTr with few output columns (N <= 64): the small-batch shape of a matrix applied to a handful of vectors. Few-row products
// (M <= 64) are computed transposed on the same kernel (gemm_f32.hip, wgk_gemm_f32), and GEMVs with 9-64 right-hand sides land here too.
//
// The 256 x 128 tile of gemm_f32.hip spends 50-87 % of its MFMAs on columns that do not exist here (4096 x 16 x 4096: 41 us against
// 18 us for the vendor's 32 x 16 tile), and the operation is not MFMA-bound anyway: every element of A is used N times, so it is
// bound by streaming A from HBM like a GEMV. The kernel below is built for that: every wave owns 32 rows for the whole K range and streams
// them -- and its own copy of the small B -- through a wave-private LDS ring by LDS-DMA: no barrier, no cross-wave sum. K is split over
// grid.y into f32 partial slabs that wg_splitk_reduce adds in ascending order (deterministic; alpha / beta / output view applied there);
// with a single chunk the kernel writes the output itself.
// History (profiles/r01_evidence.md section 14): a first kernel streamed A straight from global memory into MFMA operands (register ring,
// 4 waves splitting K, LDS sum) -- 20-40 % slower than this one once its loads overlapped the MFMAs; a shared B stage with one barrier per
// stage was 50-70 % slower (the waves drift with memory latency); a 16-wide MFMA variant for N <= 16 changed nothing THEN (round 1, a slower stream) and is
// the shipped form for N <= 16 since round 5 (W16 below): with the stream at 5.3 TB/s the 32-wide MFMAs kept the matrix pipe busy 58 % of the time and the clock at 1.9 GHz.
#include "wg_internal.hpp"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4v __attribute__((ext_vector_type(4)));
#ifndef WG_SKINNY_SWZ
#define WG_SKINNY_SWZ 0 // 1: Gemm (column-major A): the [32 k][32 m] stage image with its 16-byte chunks permuted so that the MFMA operand reads are bank-conflict-free.
                        // Built and measured in round 6 (profiles/r06_skinny_swz_ab.txt): correct, the conflicts are gone -- and the kernel is SLOWER (32000 x 16 x 4096
                        // 6430 -> 6350 GB/s, 16384 x 32 x 4096 51 -> 61 us): this kernel is bound by the HBM stream, its LDS has time to spare for a 4-way conflict, and
                        // the per-lane selects that undo the row swap make every MFMA wait for all four reads of its operand. Off; kept as the A/B build.
#endif
#ifndef WG_SKINNY_W16
#define WG_SKINNY_W16 1 // N <= 16 on v_mfma_f32_16x16x4_f32 (two 16-row sub-tiles per wave) instead of 32x32x2 with half of its columns padding (0: A/B builds)
#endif

struct SkinnyArgs {
    const float *a; uint32_t lda; uint64_t a_batch;
    const float *b; uint32_t ldb; uint64_t b_batch;
    float *part;               // slabs [z][split][N][M] (nsplit > 1)
    float *c; uint32_t ldc; uint64_t c_batch; float alpha, beta; // the output view: written directly when nsplit == 1
    uint32_t npanels;          // blockIdx.z = matrix * npanels + column panel of 32 NT columns (1: the whole of a few-column output)
    uint32_t crs;              // element stride between consecutive ROWS of the output (1; != 1: the caller wants it transposed, beta == 0)
    uint32_t M, N, K;
    uint32_t nsplit, k_per_split; // k_per_split % 32 == 0
    uint32_t rot;              // GemmTr: workgroups start their sweep over K at different stages (see the kernel)
    uint32_t a_nt;             // 1: the streamed operand's pieces carry the non-temporal hint (tr_dma_streamed)
};

__device__ __forceinline__ float comp4(const float4 &v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }

// ---------------------------------------------------------------------------------------------------------------------------------
// TRANS_A = true: GemmTr; false: Gemm.
// GemmTr with few columns (m1 stored K x M: every output row is a k-contiguous column, like the columns of m2). Lanes reading float4s
// along k of 32 different columns straight from global memory touch 32 cache lines per instruction (measured: no better than the
// tiles), so here the rows go through LDS, fetched as WHOLE 128-byte lines by LDS-DMA -- and, since every wave owns its own 32 rows for
// the whole K range, entirely wave-private: no barrier, no cross-wave sum.
//   * wave w of a workgroup owns rows 32 w .. 32 w + 31 of a 128-row block = ONE M-tile of v_mfma_f32_32x32x2_f32 (16 NT accumulators);
//   * a stage is 32 k = one line per row: 4 DMA pieces of A (8 rows x 128 B) + 4 NT of B (the wave's own copy: B is small and comes
//     from L2) into the wave's slot of a 4- (NT = 2: 3-) stage ring (4 (1 + NT) KiB per wave and stage), counted vmcnt;
//   * LDS image [row][8 chunks of 16 B], chunk c at position c ^ ((row >> 1) & 7): conflict-free ds_read_b128 for the instruction's
//     real lane groups; per 8 k, half-wave h takes chunk 2 ks + h, MFMA s its component s (the k order of gemm_f32.hip);
//   * a K % 32 remainder: the last stage's missing chunks are fetched from clamped addresses and zeroed after the LDS read.
//   * Gemm (column-major A): the wave's 32 rows at one k are one 128-byte line, a piece is 8 k, the LDS image [32 k][32 m] and the A operand
//     of an MFMA one ds_read_b32 (consecutive lanes, consecutive floats); everything else is the same.
// M0 is written without save / restore (tests/test_abi_and_host.py checks the ISA). Bound: HBM (A read once).
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t TR_BIAS = 3072; // see M16_BIAS in gemm_f16.hip
template <int IMM>
__device__ __forceinline__ void tr_dma(uint32_t voff, const void *sbase) {
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%c2" ::"v"(voff), "s"(sbase), "i"(IMM));
}
// the streamed operand's pieces: with the non-temporal hint when the matrix is read once and is too large for the 256 MiB Infinity Cache anyway (SkinnyArgs::a_nt: the
// launcher). Measured per shape (profiles/r05_skinny_nt_ab.txt): 32000 x 16 x 4096 0.734 -> 0.79-0.80 of HBM peak, Gemv 65536 x 4096 x 8 174 -> 152 us, f16 GemvTr 4096 x
// 65536 x 8 90 -> 80; matrices that fit the cache and are swept repeatedly LOSE with the hint (4096 x 11008 x 4 32 -> 36 us), hence by size. `nt` alone: the scope bits
// (sc0 / sc1) on top changed nothing.
template <int IMM>
__device__ __forceinline__ void tr_dma_streamed(uint32_t voff, const void *sbase, uint32_t nt) {
    if (nt) asm volatile("global_load_lds_dwordx4 %0, %1 offset:%c2 nt" ::"v"(voff), "s"(sbase), "i"(IMM));
    else asm volatile("global_load_lds_dwordx4 %0, %1 offset:%c2" ::"v"(voff), "s"(sbase), "i"(IMM));
}
__device__ __forceinline__ void tr_set_m0(uint32_t lds_dst) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_dst)); }

// B_KMAJ: m2 is given with its COLUMNS contiguous (element (k, n) at b[n + k * ldb]) -- the untransposed m1 of a few-row Gemm computed
// transposed; its stage image is [32 k][32 n] per column tile, read like the column-major A image.
// W16 (round 5, N <= 16): the same stages, rings and LDS images, multiplied by v_mfma_f32_16x16x4_f32 -- lane (i16 = lane & 15, kq = lane >> 4) feeds row / column i16
// with k = 4 (kq + 4 j) + s of the stage (j = 0, 1; s = the component of its float4): two 16-row sub-tiles x 8 MFMAs of 32 cycles per stage instead of 16 of 64 (the
// 32-wide tile spends half of its MFMA cycles -- 7/8 for 4 columns -- on columns that do not exist, and the counters show the MFMA pipe busy 58 % of this HBM-bound
// kernel's time at N = 16: profiles/r05_evidence.md section 5). C/D: lane (i16, kq) holds column i16, rows 4 kq .. 4 kq + 3 of each sub-tile.
// T = _Float16 (round 5; GemmTr with N <= 16 only, i.e. TRANS_A + W16 with k-contiguous m2): the same stages, rings and images at 64 k per 128-byte line; a lane's
// 16-byte chunk is the 8 halves v_mfma_f32_16x16x32_f16 takes from it (k = 8 kq .. 8 kq + 7 of a 32-k block: chunk kq + 4 j is block j), f32 accumulation, one
// rounding at the store (split partials stay f32). What it is for: f16 GemvTr with 3 .. 16 right-hand sides / f16 GemmTr with few columns, which the tiled
// f16 kernels ran at 3.9 TB/s (65536 x 4096 x 8: 138 us, vendor 115).
typedef _Float16 sk_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 sk_h4 __attribute__((ext_vector_type(4)));
template <bool TRANS_A, int NT, bool B_KMAJ = false, bool W16 = false, typename T = float>
__global__ __launch_bounds__(256, 1) void gemm_f32_skinny_kernel(SkinnyArgs g) {
    static_assert(!W16 || NT == 1, "the 16-wide form has one column tile");
    constexpr bool F16 = sizeof(T) == 2;
    static_assert(!F16 || (TRANS_A && W16 && !B_KMAJ), "f16: GemmTr, N <= 16, k-contiguous m2");
    constexpr uint32_t ES = sizeof(T), KS = 128u / ES, CK = 16u / ES; // element size; k per stage (one 128-byte line per row); k per 16-byte chunk
    // (W16 stages the same 32 columns of B and keeps the 4-stage ring: staging only the 16 it uses and giving the room to a 6-stage ring -- 20 KiB of A in flight per
    // wave instead of 12 -- measured slower: 32000 x 16 x 4096 5.75 -> 5.57 TB/s, GemvTr 8192^2 x 3 43.1 -> 46.5 us; and so did a 3-stage ring of 6 KiB stages with TWO
    // workgroups per CU (72 KiB each): 5.73 -> 5.29 TB/s; profiles/r05_evidence.md section 5)
    constexpr int BP = 4 * NT;                        // DMA pieces of B per wave and stage
    constexpr int RING = NT == 1 ? 4 : 3;             // 128 / 144 KiB of LDS: one workgroup per CU
    constexpr int STAGE_BYTES = 4096 + 1024 * BP;     // per wave: A 32 rows x 128 B, then B 32 NT (16) columns x 128 B
    constexpr int PIECES = 4 + BP;                    // DMA pieces per wave and stage
    __shared__ __attribute__((aligned(16))) char smem[4 * RING * STAGE_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const uint32_t r0 = blockIdx.x * 128u + 32u * wave;
    const uint32_t z = blockIdx.z / g.npanels, col0 = (blockIdx.z % g.npanels) * (32u * NT), split = blockIdx.y;
    const uint32_t kb = split * g.k_per_split;             // multiple of KS
    const uint32_t ke = min(kb + g.k_per_split, g.K);
    if (kb >= ke) return;                                   // (the launcher leaves no empty split)
    const uint32_t nst = (ke - kb + KS - 1u) / KS;
    const uint32_t last_chunks = (ke - kb - KS * (nst - 1u)) / CK; // valid 16-byte chunks of the last stage: 1 .. 8

    // ---- DMA addressing: piece q = rows 8 q .. 8 q + 7 of the wave's 32 (lane -> row 8 q + (lane >> 3), position lane & 7) ----
    const T *A = reinterpret_cast<const T *>(g.a) + z * g.a_batch + (TRANS_A ? (uint64_t)kb : (uint64_t)kb * g.lda);
    const T *B = reinterpret_cast<const T *>(g.b) + z * g.b_batch + (B_KMAJ ? (uint64_t)kb * g.ldb : (uint64_t)kb);
    uint32_t a_voff[4], b_voff[BP], a_tail[4], b_tail[BP]; // byte offsets; *_tail: the last stage, k clamped into the matrix
    const uint32_t kmax = g.K - CK - kb - KS * (nst - 1u);         // largest valid k offset (in elements) of a chunk in the last stage
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t rl = 8u * q + (lane >> 3);
        if constexpr (TRANS_A) { // piece q = rows 8 q .. 8 q + 7, one 128-byte line (32 k) each
            const uint32_t chunk = (lane & 7u) ^ ((rl >> 1) & 7u);
            const uint64_t row = min(r0 + rl, g.M - 1u);
            // 32-bit offsets relative to the wave's first row keep the per-lane address in one VGPR (the launcher checks the range)
            const uint32_t rel = (uint32_t)((row - min(r0, g.M - 1u)) * g.lda) * ES;
            a_voff[q] = rel + 16u * chunk + (TR_BIAS - 1024u * q);
            a_tail[q] = rel + ES * min(CK * chunk, kmax) + (TR_BIAS - 1024u * q);
        } else { // column-major A: piece q = k rows 8 q .. 8 q + 7 of the stage, the wave's 32 rows = one 128-byte line each: image [32 k][32 m]
            // Bank conflicts (round 5's counters: 4.1 M per launch at 32000 x 16 x 4096): an operand read takes ONE float per lane at k = 4 c + s, and rows 4 (or 8, 12) k
            // apart start on the same bank -- the four lane groups of the 16-wide form met on 16 banks (4-way), the two half-waves of the 32-wide form on 32 (2-way).
            // A DMA lane may fetch any 16-byte chunk, so the image is permuted instead: logical (k, chunk c = m / 4) sits at row (k & 7) ^ ((k >> 2) & 1) of its piece,
            // chunk position c ^ 4 ((k >> 3) & 1) -- rows 4 apart land on opposite bank halves, rows 8 apart on opposite halves of the row (the reads: `compute`).
            const uint32_t kr = WG_SKINNY_SWZ ? 8u * q + ((lane >> 3) ^ ((lane >> 5) & 1u)) : rl;
            const uint32_t ch = WG_SKINNY_SWZ ? (lane & 7u) ^ (4u * ((uint32_t)q & 1u)) : (lane & 7u);
            const uint32_t m = min(r0 + 4u * ch, g.M - 4u) - min(r0, g.M - 4u); // M % 4 == 0
            a_voff[q] = (kr * g.lda + m) * 4u + (TR_BIAS - 1024u * q);
            a_tail[q] = (min(kr, kmax + 3u) * g.lda + m) * 4u + (TR_BIAS - 1024u * q);
        }
    }
#pragma unroll
    for (int q = 0; q < BP; ++q) {
        if constexpr (B_KMAJ) { // piece q & 3 = k rows 8 (q & 3) .. + 7 of column tile q >> 2; 4 columns per lane (N % 4 == 0)
            const uint32_t kr = 8u * (q & 3) + (lane >> 3);
            const uint32_t col = min(col0 + 32u * (q >> 2) + 4u * (lane & 7u), g.N - 4u);
            b_voff[q] = (kr * g.ldb + col) * 4u + (TR_BIAS - 1024u * (q & 3));
            b_tail[q] = (min(kr, kmax + 3u) * g.ldb + col) * 4u + (TR_BIAS - 1024u * (q & 3));
            continue;
        }
        const uint32_t cl = 8u * q + (lane >> 3);
        const uint32_t chunk = (lane & 7u) ^ ((cl >> 1) & 7u);
        const uint32_t col = min(cl, g.N - 1u - col0); // relative to the panel's first column (gb0 below)
        b_voff[q] = col * g.ldb * ES + 16u * chunk + (TR_BIAS - 1024u * (q & 3));
        b_tail[q] = col * g.ldb * ES + ES * min(CK * chunk, kmax) + (TR_BIAS - 1024u * (q & 3));
    }
    const uint64_t b_step = B_KMAJ ? (uint64_t)128u * g.ldb : 128u; // bytes per stage (32 k)
    const char *ga0 = (const char *)(TRANS_A ? A + (uint64_t)min(r0, g.M - 1u) * g.lda : A + min(r0, g.M - 4u)) - TR_BIAS;
    const uint64_t a_step = TRANS_A ? 128u : (uint64_t)128u * g.lda; // bytes per stage (32 k)
    const char *gb0 = (const char *)(B_KMAJ ? B : B + (uint64_t)col0 * g.ldb) - TR_BIAS;
    const uint32_t lds_wave = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem + wave * (RING * STAGE_BYTES));
#ifndef WG_SKINNY_ROT
#define WG_SKINNY_ROT 1
#endif
    // GemmTr, whole stages only: every workgroup starts its sweep over K at its own stage and wraps around. All waves of the chip walk k at about the
    // same pace; with k-contiguous rows a fixed number of KiB apart, "the same k of every row" is the same few memory channels for everybody
    // (65536 x 8 x 4096 GemmTr: 3.8 TB/s; column-major A does not have this: its workgroups differ in the LOW address bits). Trip j of a wave works
    // on stage (j + rot) mod nst; the sum over k is the same set of products in a rotated order (deterministic per workgroup). Only while m2's K range is
    // small (<= 256 KiB, the launcher's g.rot): every workgroup re-reads it from L2, and in step they all read the same lines of it -- out of step a
    // larger m2 stops being a broadcast (N = 64, K = 4096: 325 -> 423 us with the rotation; N = 8: 281 -> 228).
    const uint32_t a_nt = g.a_nt; // (workgroup-uniform: a scalar branch around each of the four pieces)
    const uint32_t rot = (WG_SKINNY_ROT && TRANS_A && g.rot && last_chunks == 8u && nst > 1u) ? (uint32_t)(((uint64_t)blockIdx.x * 7u + blockIdx.y * 3u) % nst) : 0u; // (per workgroup: its four waves fetch the same stage of the small m2 together)
    auto issue = [&](uint32_t trip) { // trip -> ring slot trip % RING; the stage it carries: (trip + rot) mod nst
        const uint32_t dst = lds_wave + (trip % RING) * STAGE_BYTES;
        uint32_t st = trip + rot;
        if (st >= nst) st -= nst;
        const char *ga = ga0 + (uint64_t)st * a_step, *gb = gb0 + (uint64_t)st * b_step;
        const bool tail = rot == 0u && st + 1u == nst; // wave-uniform (with a rotation there is no partial stage)
        tr_set_m0(dst);
        tr_dma_streamed<0>(tail ? a_tail[0] : a_voff[0], ga, a_nt); tr_dma_streamed<1024>(tail ? a_tail[1] : a_voff[1], ga, a_nt);
        tr_dma_streamed<2048>(tail ? a_tail[2] : a_voff[2], ga, a_nt); tr_dma_streamed<3072>(tail ? a_tail[3] : a_voff[3], ga, a_nt);
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            tr_set_m0(dst + 4096u * (1 + u));
            tr_dma<0>(tail ? b_tail[4 * u + 0] : b_voff[4 * u + 0], gb); tr_dma<1024>(tail ? b_tail[4 * u + 1] : b_voff[4 * u + 1], gb);
            tr_dma<2048>(tail ? b_tail[4 * u + 2] : b_voff[4 * u + 2], gb); tr_dma<3072>(tail ? b_tail[4 * u + 3] : b_voff[4 * u + 3], gb);
        }
    };

    floatx16 acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[u][e] = 0.f;
    floatx4v acc16[2] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } }; // W16: sub-tile t = rows 16 t .. 16 t + 15 of the wave's 32
    const int i16 = lane & 15, kq = lane >> 4;

    const uint32_t rd = (uint32_t)i * 128u, sw = (uint32_t)((i >> 1) & 7);
    const char *wbase = smem + wave * (RING * STAGE_BYTES);
    for (uint32_t st = 0; st < (uint32_t)(RING - 1) && st < nst; ++st) issue(st);
    // one stage's MFMAs on ring slot sl. TAIL (the last stage only): chunks at or past `valid` lie outside the K range -- they were fetched from
    // clamped addresses and are zeroed here. The steady-state instance has no selects: as `live ? load : 0` hipcc sinks the LDS reads into
    // branches, which costs the overlap of the next substep's reads with this substep's MFMAs.
    auto compute = [&](const char *sl, auto tail_c, uint32_t valid) {
        constexpr bool TAIL = decltype(tail_c)::value;
        if constexpr (W16 && F16) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t c = (uint32_t)kq + 4u * j; // this lane's 16-byte chunk of the stage: k = 8 c .. 8 c + 7 = the lane's share of 32-k block j
                sk_h8 ah[2], bh;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const uint32_t row = 16u * t + (uint32_t)i16;
                    ah[t] = *reinterpret_cast<const sk_h8 *>(sl + row * 128u + ((c ^ ((row >> 1) & 7u)) * 16u));
                }
                bh = *reinterpret_cast<const sk_h8 *>(sl + 4096u + (uint32_t)i16 * 128u + ((c ^ (((uint32_t)i16 >> 1) & 7u)) * 16u));
                if constexpr (TAIL) {
                    if (c >= valid) { // (registers already loaded; the zeroing is a select per dword)
                        const sk_h8 zero = { 0, 0, 0, 0, 0, 0, 0, 0 };
                        ah[0] = zero; ah[1] = zero; bh = zero;
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) acc16[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc16[t], 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            return;
        } else if constexpr (W16) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t c = (uint32_t)kq + 4u * j; // this lane's 16-byte chunk of the stage: k = 4 c .. 4 c + 3
                float4 af[2], bf;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const uint32_t row = 16u * t + (uint32_t)i16;
                    if constexpr (TRANS_A) af[t] = *reinterpret_cast<const float4 *>(sl + row * 128u + ((c ^ ((row >> 1) & 7u)) * 16u));
                    else if constexpr (WG_SKINNY_SWZ) { // k = 4 c + s sits at row 4 c + (s ^ (c & 1)), m at m ^ 16 ((c >> 1) & 1): (c & 1, (c >> 1) & 1) = (kq & 1, kq >> 1), per lane
                        const float *ak = reinterpret_cast<const float *>(sl + (4u * c) * 128u) + (row ^ (16u * (((uint32_t)kq >> 1) & 1u)));
                        const float r0_ = ak[0], r1_ = ak[32], r2_ = ak[64], r3_ = ak[96];
                        const bool odd = (kq & 1) != 0;
                        af[t] = make_float4(odd ? r1_ : r0_, odd ? r0_ : r1_, odd ? r3_ : r2_, odd ? r2_ : r3_);
                    } else {
                        const float *ak = reinterpret_cast<const float *>(sl + (4u * c) * 128u) + row;
                        af[t] = make_float4(ak[0], ak[32], ak[64], ak[96]);
                    }
                }
                if constexpr (B_KMAJ) {
                    const float *bk = reinterpret_cast<const float *>(sl + 4096u + (4u * c) * 128u) + i16;
                    bf = make_float4(bk[0], bk[32], bk[64], bk[96]);
                } else bf = *reinterpret_cast<const float4 *>(sl + 4096u + (uint32_t)i16 * 128u + ((c ^ (((uint32_t)i16 >> 1) & 7u)) * 16u));
                if constexpr (TAIL) {
                    const bool live = c < valid;
                    asm volatile("" : "+v"(af[0].x), "+v"(af[0].y), "+v"(af[0].z), "+v"(af[0].w), "+v"(af[1].x), "+v"(af[1].y), "+v"(af[1].z), "+v"(af[1].w));
                    asm volatile("" : "+v"(bf.x), "+v"(bf.y), "+v"(bf.z), "+v"(bf.w));
                    af[0] = live ? af[0] : make_float4(0.f, 0.f, 0.f, 0.f);
                    af[1] = live ? af[1] : make_float4(0.f, 0.f, 0.f, 0.f);
                    bf = live ? bf : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc16[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4(af[t], sidx), comp4(bf, sidx), acc16[t], 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint32_t c = 2u * ks + h;
            float4 af;
            if constexpr (TRANS_A) af = *reinterpret_cast<const float4 *>(sl + rd + ((c ^ sw) * 16u));
            else if constexpr (WG_SKINNY_SWZ) { // [k][32 m], permuted (see the DMA offsets): c = 2 ks + h, so row 4 c + (s ^ h) and m ^ 16 (ks & 1)
                const float *ak = reinterpret_cast<const float *>(sl + (4u * c) * 128u) + ((uint32_t)i ^ (16u * ((uint32_t)ks & 1u)));
                const float r0_ = ak[0], r1_ = ak[32], r2_ = ak[64], r3_ = ak[96];
                const bool odd = h != 0;
                af = make_float4(odd ? r1_ : r0_, odd ? r0_ : r1_, odd ? r3_ : r2_, odd ? r2_ : r3_);
            } else { // [k][32 m]: k = 4 c + s, one float per MFMA
                const float *ak = reinterpret_cast<const float *>(sl + (4u * c) * 128u) + i;
                af = make_float4(ak[0], ak[32], ak[64], ak[96]);
            }
            float4 bf[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                if constexpr (B_KMAJ) {
                    const float *bk = reinterpret_cast<const float *>(sl + 4096u * (1 + u) + (4u * c) * 128u) + i;
                    bf[u] = make_float4(bk[0], bk[32], bk[64], bk[96]);
                } else
                bf[u] = *reinterpret_cast<const float4 *>(sl + 4096u * (1 + u) + rd + ((c ^ sw) * 16u));
            }
            if constexpr (TAIL) {
                const bool live = c < valid; // select on registers that are already loaded (the empty asm keeps the loads out of a branch)
                asm volatile("" : "+v"(af.x), "+v"(af.y), "+v"(af.z), "+v"(af.w));
                af = live ? af : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    asm volatile("" : "+v"(bf[u].x), "+v"(bf[u].y), "+v"(bf[u].z), "+v"(bf[u].w));
                    bf[u] = live ? bf[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp4(af, sidx), comp4(bf[u], sidx), acc[u], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): this slot's reads are done before a later trip's DMA refills it
    };
    uint32_t st = 0;
    for (; st + (uint32_t)(RING - 1) < nst; ++st) { // steady state: stage st + RING - 1 exists, so stage st is never the last
        // slot (st - 1) % RING was read in the previous trip: refill it with stage st + RING - 1, then wait for stage st
        issue(st + RING - 1);
        asm volatile("s_waitcnt vmcnt(%c0)" ::"i"((RING - 1) * PIECES) : "memory");
        compute(wbase + (st % RING) * STAGE_BYTES, std::false_type{}, 8u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the last <= RING - 1 stages: nothing more is issued
    for (; st + 1u < nst; ++st) compute(wbase + (st % RING) * STAGE_BYTES, std::false_type{}, 8u);
    compute(wbase + (st % RING) * STAGE_BYTES, std::true_type{}, last_chunks);

    // C/D map: lane (i, h), register e -> row (e&3) + 8 (e>>2) + 4 h of the wave's 32, column i: registers 4 gq .. 4 gq + 3 = 4 consecutive rows
    const bool direct = g.nsplit == 1;
    float *P = (direct && !F16) ? g.c + z * g.c_batch : g.part + ((uint64_t)z * g.nsplit + split) * ((uint64_t)g.M * g.N);
    const uint32_t ldp = direct ? g.ldc : g.M;
    auto put4 = [&](uint32_t col, uint32_t row, float4 v) { // rows row .. row + 3 of column col (M % 4 == 0: all in or all out)
        float *dst = P + (uint64_t)col * ldp + row; // (any element-aligned address: wg_internal.hpp wg_ld_u / wg_st_u)
        if (direct) {
            if (g.alpha != 1.f) { v.x *= g.alpha; v.y *= g.alpha; v.z *= g.alpha; v.w *= g.alpha; }
            if (g.crs != 1u) { // transposed output (few-row products): four scalars, rows g.crs apart
                float *t = P + (uint64_t)col * ldp + (uint64_t)row * g.crs;
                t[0] = v.x; t[g.crs] = v.y; t[2u * g.crs] = v.z; t[3u * g.crs] = v.w;
                return;
            }
            if (g.beta != 0.f) {
                const float4 o = wg_ld_u(dst);
                v.x = fmaf(g.beta, o.x, v.x); v.y = fmaf(g.beta, o.y, v.y); v.z = fmaf(g.beta, o.z, v.z); v.w = fmaf(g.beta, o.w, v.w);
            }
        }
        wg_st_u(dst, v);
    };
    if constexpr (W16) {
        const uint32_t col = col0 + (uint32_t)i16;
        if (col < g.N) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t row = r0 + 16u * t + 4u * (uint32_t)kq;
                if (row >= g.M) continue;
                float4 v = make_float4(acc16[t][0], acc16[t][1], acc16[t][2], acc16[t][3]);
                if constexpr (F16) {
                    if (!direct) { put4(col, row, v); continue; } // f32 partial slab
                    _Float16 *dst = reinterpret_cast<_Float16 *>(g.c) + z * g.c_batch + (uint64_t)col * g.ldc + row;
                    if (g.alpha != 1.f) { v.x *= g.alpha; v.y *= g.alpha; v.z *= g.alpha; v.w *= g.alpha; }
                    if (g.beta != 0.f) {
                        const sk_h4 o = *reinterpret_cast<const sk_h4 *>(dst);
                        v.x = fmaf(g.beta, (float)o[0], v.x); v.y = fmaf(g.beta, (float)o[1], v.y); v.z = fmaf(g.beta, (float)o[2], v.z); v.w = fmaf(g.beta, (float)o[3], v.w);
                    }
                    const sk_h4 hv = { (_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w };
                    *reinterpret_cast<sk_h4 *>(dst) = hv;
                } else put4(col, row, v);
            }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const uint32_t col = col0 + 32u * u + i;
        if (col >= g.N) continue;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const uint32_t row = r0 + 8u * gq + 4u * h;
            if (row >= g.M) continue; // M % 4 == 0
            float4 v = make_float4(acc[u][4 * gq + 0], acc[u][4 * gq + 1], acc[u][4 * gq + 2], acc[u][4 * gq + 3]);
            float *dst = P + (uint64_t)col * ldp + row; // (any element-aligned address: wg_internal.hpp wg_ld_u / wg_st_u)
            if (direct) {
                if (g.alpha != 1.f) { v.x *= g.alpha; v.y *= g.alpha; v.z *= g.alpha; v.w *= g.alpha; }
                if (g.crs != 1u) { // transposed output (few-row products): four scalars, rows g.crs apart
                    float *t = P + (uint64_t)col * ldp + (uint64_t)row * g.crs;
                    t[0] = v.x; t[g.crs] = v.y; t[2u * g.crs] = v.z; t[3u * g.crs] = v.w;
                    continue;
                }
                if (g.beta != 0.f) {
                    const float4 o = wg_ld_u(dst);
                    v.x = fmaf(g.beta, o.x, v.x); v.y = fmaf(g.beta, o.y, v.y); v.z = fmaf(g.beta, o.z, v.z); v.w = fmaf(g.beta, o.w, v.w);
                }
            }
            wg_st_u(dst, v);
        }
    }
}

} // namespace

// out = alpha * m1 * m2 + beta * out for N <= 64 (NN only). Returns WG_ERR_UNSUPPORTED-free: the caller checks applicability.
int wgk_gemm_f32_skinny(wg_ctx *ctx, bool trans, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats, float *out, uint32_t out_ld, uint64_t out_batch,
                        wgk_mat m1, wgk_mat m2, float alpha, float beta, uint32_t out_row_stride, bool m2_kmajor, uint32_t ns_force) {
    const int cus = ctx->compute_units > 0 ? ctx->compute_units : 256;
    const uint32_t row_blocks = (M + 127u) / 128u;
    // K splits: the count whose workgroups fill whole rounds of the CUs with the least k per round (11008 rows = 86 row blocks: 3 splits
    // = 258 workgroups would run a second round for two of them; 5 splits = 430 run two rounds of 820 k). Measured: one long workgroup
    // per CU beats several short ones (4096 x 16 x 4096: 24 us with 256 workgroups, 32 us with 1024), so ties go to fewer splits and
    // every extra split is charged the k-equivalent of its slab + epilogue.
    const uint32_t max_split = (K + 127u) / 128u; // >= 128 k per workgroup
    // more than 64 columns (small squares, see wgk_gemm_f32): 64-column panels over grid.z, every panel streaming A from L2
    const uint32_t npanels = N > 64u ? (N + 63u) / 64u : 1u;
    if ((uint64_t)nmats * npanels > 65535u) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: too many matrices x column panels");
    const uint64_t blocks = (uint64_t)row_blocks * nmats * npanels;
    uint32_t ns = 1;
    uint64_t best = ~0ull;
    for (uint32_t c = 1; c <= max_split && (uint64_t)c * blocks <= 4ull * cus + blocks; ++c) {
        if ((uint64_t)c * M * N * nmats * 4u > (512ull << 20)) break;
        const uint64_t rounds = (blocks * c + cus - 1) / cus;
        const uint64_t cost = rounds * ((K + c - 1) / c + 128u); // + pipeline fill, epilogue and slab per round (11008 x 32 x 4096: 5 splits 49 us, 11 splits 52)
        if (cost < best) { best = cost; ns = c; }
    }
    if (ns_force) ns = ns_force > max_split ? max_split : ns_force;
    uint32_t kps = (((K + ns - 1) / ns) + 31u) & ~31u;
    ns = (K + kps - 1) / kps;
    if (ns > 65535u || nmats > 65535u) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: too many splits or matrices for the skinny path");
    void *ws = nullptr;
    if (ns > 1)
        if (int rc = wg_ctx_workspace(ctx, (size_t)ns * M * N * nmats * sizeof(float), &ws)) return rc;
    SkinnyArgs g;
    g.c = out; g.ldc = out_ld; g.c_batch = out_batch; g.alpha = alpha; g.beta = beta; g.crs = out_row_stride;
    g.a = (const float *)m1.ptr; g.lda = m1.ld; g.a_batch = m1.batch;
    g.b = (const float *)m2.ptr; g.ldb = m2.ld; g.b_batch = m2.batch;
    g.part = (float *)ws; g.M = M; g.N = N; g.K = K; g.nsplit = ns; g.k_per_split = kps; g.npanels = npanels;
    g.rot = (uint64_t)(ns > 1 ? kps : K) * (N < 32u * (N <= 32 ? 1u : 2u) ? N : 32u * (N <= 32 ? 1u : 2u)) * 4u <= (256u << 10) ? 1u : 0u;
    g.a_nt = (npanels == 1 && (uint64_t)M * K * 4u >= (384ull << 20)) ? 1u : 0u; // read once, and no use to the 256 MiB Infinity Cache (tr_dma_streamed)
    const dim3 grid(row_blocks, ns, nmats * npanels);
    const bool w16 = WG_SKINNY_W16 && N <= 16u && npanels == 1u;
    if (m2_kmajor) { // GemmTr only (the few-row route)
        if (w16) hipLaunchKernelGGL((gemm_f32_skinny_kernel<true, 1, true, true>), grid, dim3(256), 0, ctx->stream, g);
        else if (N <= 32) hipLaunchKernelGGL((gemm_f32_skinny_kernel<true, 1, true>), grid, dim3(256), 0, ctx->stream, g);
        else hipLaunchKernelGGL((gemm_f32_skinny_kernel<true, 2, true>), grid, dim3(256), 0, ctx->stream, g);
    } else if (trans) {
        if (w16) hipLaunchKernelGGL((gemm_f32_skinny_kernel<true, 1, false, true>), grid, dim3(256), 0, ctx->stream, g);
        else if (N <= 32) hipLaunchKernelGGL((gemm_f32_skinny_kernel<true, 1>), grid, dim3(256), 0, ctx->stream, g);
        else hipLaunchKernelGGL((gemm_f32_skinny_kernel<true, 2>), grid, dim3(256), 0, ctx->stream, g);
    } else if (w16) hipLaunchKernelGGL((gemm_f32_skinny_kernel<false, 1, false, true>), grid, dim3(256), 0, ctx->stream, g);
    else if (N <= 32) hipLaunchKernelGGL((gemm_f32_skinny_kernel<false, 1>), grid, dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL((gemm_f32_skinny_kernel<false, 2>), grid, dim3(256), 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    if (ns == 1) return WG_OK;
    if (out_row_stride != 1u) return wg_splitk_reduce_strided(ctx, g.part, ns, M, N, nmats, out, out_row_stride, out_ld, out_batch, alpha);
    return wg_splitk_reduce(ctx, g.part, ns, M, N, nmats, WG_F32, out, out_ld, out_batch, alpha, beta);
}

// f16 GemmTr with N <= 16 (f16 GemvTr with a few right-hand sides arrives here through wgk_gemm_f16): out = alpha * m1^T * m2 + beta * out, m1 stored K x M and m2
// K x N, both k-contiguous. The caller (wgk_gemm_f16) has checked: K % 8 == 0, leading dimensions % 8 == 0, 16-byte aligned bases, 32-bit offsets in range.
int wgk_gemm_f16_skinny(wg_ctx *ctx, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats, __half *out, uint32_t out_ld, uint64_t out_batch, wgk_mat m1, wgk_mat m2,
                        float alpha, float beta) {
    const int cus = ctx->compute_units > 0 ? ctx->compute_units : 256;
    const uint32_t row_blocks = (M + 127u) / 128u;
    const uint32_t max_split = (K + 255u) / 256u; // >= 256 k (4 stages) per workgroup
    const uint64_t blocks = (uint64_t)row_blocks * nmats;
    uint32_t ns = 1;
    uint64_t best = ~0ull;
    for (uint32_t c = 1; c <= max_split && (uint64_t)c * blocks <= 4ull * cus + blocks; ++c) { // the f32 launcher's plan at 64 k per stage
        if ((uint64_t)c * M * N * nmats * 4u > (512ull << 20)) break;
        const uint64_t rounds = (blocks * c + cus - 1) / cus;
        const uint64_t cost = rounds * ((K + c - 1) / c + 256u);
        if (cost < best) { best = cost; ns = c; }
    }
    uint32_t kps = (((K + ns - 1) / ns) + 63u) & ~63u;
    ns = (K + kps - 1) / kps;
    if (ns > 65535u || nmats > 65535u) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: too many splits or matrices for the skinny path");
    void *ws = nullptr;
    if (ns > 1)
        if (int rc = wg_ctx_workspace(ctx, (size_t)ns * M * N * nmats * sizeof(float), &ws)) return rc;
    SkinnyArgs g;
    g.c = (float *)out; g.ldc = out_ld; g.c_batch = out_batch; g.alpha = alpha; g.beta = beta; g.crs = 1;
    g.a = (const float *)m1.ptr; g.lda = m1.ld; g.a_batch = m1.batch;
    g.b = (const float *)m2.ptr; g.ldb = m2.ld; g.b_batch = m2.batch;
    g.part = (float *)ws; g.M = M; g.N = N; g.K = K; g.nsplit = ns; g.k_per_split = kps; g.npanels = 1;
    g.rot = (uint64_t)(ns > 1 ? kps : K) * N * 2u <= (256u << 10) ? 1u : 0u;
    g.a_nt = (uint64_t)M * K * 2u >= (384ull << 20) ? 1u : 0u;
    hipLaunchKernelGGL((gemm_f32_skinny_kernel<true, 1, false, true, _Float16>), dim3(row_blocks, ns, nmats), dim3(256), 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    if (ns == 1) return WG_OK;
    return wg_splitk_reduce(ctx, g.part, ns, M, N, nmats, WG_F16, out, out_ld, out_batch, alpha, beta);
}
