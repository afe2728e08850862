// f16 Gemm: the generic any-shape fallback (64 x 64 tiles, f32 FMA on the vector units, ~40 TFLOP/s) for products too small to be worth
// staging into the MFMA kernels' shapes (gemm_f16.hip: < 2^24 multiply-adds). The previous-generation 32x32x16 MFMA kernel that used to
// live here (K % 64 != 0, short splits) is gone: the 16x16x32 kernel takes any K % 8 == 0 now (its K remainder is the accumulators'
// initial value, gemm_f16.hip m16_tile), and shorter K runs on the 128 x 128 kernel or on zero-padded copies.
#include "gemm_f16_common.hpp"

namespace wgf16 {
namespace {

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

int generic_launch(wg_ctx *ctx, bool trans, dim3 grid, const GemmArgs &g) {
    if (trans) hipLaunchKernelGGL(gemm_f16_generic_kernel<true>, grid, dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL(gemm_f16_generic_kernel<false>, grid, dim3(256), 0, ctx->stream, g);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

} // namespace wgf16
