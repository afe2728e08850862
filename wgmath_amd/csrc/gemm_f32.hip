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
// Pipeline: global -> registers (next k-tile, issued before the MFMAs) -> LDS (other buffer) -> one barrier per k-tile.
// With 2 waves per SIMD the second workgroup's MFMAs cover the first one's barrier / LDS refill.
// Edges: every global access is predicated at float4 granularity (rows/cols/K are multiples of 4 by the vec4
// precondition), out-of-range operands are zero-filled, so any M, N, K % 4 == 0 is accepted.
// Workgroup ids are remapped so that each XCD (private L2) works on a contiguous band of tiles.
#include "wg_internal.hpp"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

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
};

__device__ __forceinline__ float4 ldg4(const float *p, bool ok) {
    return ok ? *reinterpret_cast<const float4 *>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float comp(const float4 &v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }

// bijective "XCD-contiguous" remap of the linear workgroup id (guide T1): hardware deals ids round-robin to the 8 XCDs
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg) {
    const uint32_t q = nwg / 8u, r = nwg % 8u, xcd = bid % 8u, local = bid / 8u;
    const uint32_t base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + local;
}

template <bool TRANS_A>
__global__ __launch_bounds__(kThreads, 2) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][A_TILE];
    __shared__ __attribute__((aligned(16))) float Bs[2][B_TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;

    const uint32_t tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
    const uint32_t tm = tile % g.tiles_m, tn = tile / g.tiles_m;
    const uint32_t m0 = tm * BM, n0 = tn * BN;
    const uint32_t z = blockIdx.y;
    const float *A = g.a + z * g.a_batch;
    const float *B = g.b + z * g.b_batch;
    float *C = g.c + z * g.c_batch;

    floatx16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

    float4 ra[4], rb[2];

    auto load_tile = [&](uint32_t k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = tid + kThreads * r;
            if constexpr (!TRANS_A) { // A[m][k] at a + k*lda + m : float4 along m
                const uint32_t m = m0 + 4u * (f & 63), k = k0 + (f >> 6);
                ra[r] = ldg4(A + (uint64_t)k * g.lda + m, m < g.M && k < g.K);
            } else { // op(A)[m][k] at a + m*lda + k : float4 along k
                const uint32_t m = m0 + (f >> 2), k = k0 + 4u * (f & 3);
                ra[r] = ldg4(A + (uint64_t)m * g.lda + k, m < g.M && k < g.K);
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int f = tid + kThreads * r;
            const uint32_t n = n0 + (f >> 2), k = k0 + 4u * (f & 3);
            rb[r] = ldg4(B + (uint64_t)n * g.ldb + k, n < g.N && k < g.K);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = tid + kThreads * r;
            if constexpr (!TRANS_A) {
                *reinterpret_cast<float4 *>(&As[buf][(f >> 6) * BM + 4 * (f & 63)]) = ra[r];
            } else {
                const int mm = f >> 2, ch = f & 3;
                *reinterpret_cast<float4 *>(&As[buf][mm * BK + 4 * (ch ^ ((mm >> 2) & 3))]) = ra[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int f = tid + kThreads * r;
            const int nn = f >> 2, ch = f & 3;
            *reinterpret_cast<float4 *>(&Bs[buf][nn * BK + 4 * (ch ^ ((nn >> 2) & 3))]) = rb[r];
        }
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = 2 * ks + h; // this half-wave's 4 consecutive k within the 16-deep tile
            float4 bf[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int nn = wn * 64 + 32 * u + i;
                bf[u] = *reinterpret_cast<const float4 *>(&Bs[buf][nn * BK + 4 * (chunk ^ ((nn >> 2) & 3))]);
            }
            if constexpr (!TRANS_A) {
                float4 af[4]; // af[s]: rows 4i..4i+3 at k = 4*chunk + s
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    af[s] = *reinterpret_cast<const float4 *>(&As[buf][(4 * chunk + s) * BM + wm * 128 + 4 * i]);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(af[s], t), comp(bf[u], s), acc[t][u], 0, 0, 0);
            } else {
                float4 af[4]; // af[t]: row 32t + i, k = 4*chunk .. +3
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int mm = wm * 128 + 32 * t + i;
                    af[t] = *reinterpret_cast<const float4 *>(&As[buf][mm * BK + 4 * (chunk ^ ((mm >> 2) & 3))]);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(af[t], s), comp(bf[u], s), acc[t][u], 0, 0, 0);
            }
        }
    };

    const uint32_t nk = (g.K + BK - 1) / BK;
    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (uint32_t kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) load_tile((kt + 1) * BK);
        compute(buf);
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue. C/D map of the 32x32 MFMA: lane l, register e -> row (e&3) + 8*(e>>2) + 4*(l>>5), col l&31.
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t col = n0 + wn * 64 + 32 * u + i;
        if (col >= g.N) continue;
        float *cc = C + (uint64_t)col * g.ldc;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) { // e >> 2
            if constexpr (!TRANS_A) {
                // M-tile t holds rows 4*row_mfma + t: (t, e&3) enumerate 16 consecutive rows from 32*gq + 16*h
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t row = m0 + wm * 128 + 32 * gq + 16 * h + 4 * q;
                    if (row < g.M)
                        *reinterpret_cast<float4 *>(cc + row) =
                            make_float4(acc[0][u][4 * gq + q], acc[1][u][4 * gq + q], acc[2][u][4 * gq + q], acc[3][u][4 * gq + q]);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t row = m0 + wm * 128 + 32 * t + 8 * gq + 4 * h;
                    if (row < g.M)
                        *reinterpret_cast<float4 *>(cc + row) =
                            make_float4(acc[t][u][4 * gq + 0], acc[t][u][4 * gq + 1], acc[t][u][4 * gq + 2], acc[t][u][4 * gq + 3]);
                }
            }
        }
    }
}

} // namespace

int wgk_gemm_f32(wg_ctx *ctx, bool trans, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats,
                 float *out, uint32_t out_ld, uint64_t out_batch, wgk_mat m1, wgk_mat m2) {
    if (M == 0 || N == 0 || nmats == 0) return WG_OK;
    if (nmats > 65535) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: more than 65535 matrices in one call");
    GemmArgs g;
    g.a = (const float *)m1.ptr; g.lda = m1.ld; g.a_batch = m1.batch;
    g.b = (const float *)m2.ptr; g.ldb = m2.ld; g.b_batch = m2.batch;
    g.c = out; g.ldc = out_ld; g.c_batch = out_batch;
    g.M = M; g.N = N; g.K = K;
    g.tiles_m = (M + BM - 1) / BM;
    g.tiles_n = (N + BN - 1) / BN;
    const uint64_t tiles = (uint64_t)g.tiles_m * g.tiles_n;
    if (tiles > 0x7fffffffull) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: too many tiles");
    const dim3 grid((uint32_t)tiles, nmats), block(kThreads);
    if (trans) hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, block, 0, ctx->stream, g);
    else hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, block, 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}
