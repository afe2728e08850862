// Split-K support shared by the two GEMM kernels.
// A GEMM whose output has too few tiles to fill 256 CUs (small M x N, long K) is cut along K: workgroup (tile, s) computes the
// partial product over K-range s into an f32 slab of the context workspace ([z][s][N][M], dense column-major), and this
// kernel adds the slabs in ASCENDING s (deterministic; no atomics) and writes the result in the output's dtype and layout.
#include "wg_internal.hpp"

namespace {

template <typename OUT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ part, uint32_t nsplit, uint32_t M, uint32_t N,
                                                            OUT *__restrict__ out, uint32_t ldc, uint64_t c_batch, float alpha, float beta) {
    // one float4 (4 consecutive rows, M % 4 == 0) of the dense slabs per thread and trip, numbered through the whole matrix: a 64-row output keeps every lane
    // busy (one block per column left 240 of 256 threads idle: 64 x 4096 in four parts 4.8 us, the kernel's launch and one memory round trip now)
    const uint32_t z = blockIdx.z, m4s = M / 4u;
    const uint64_t slab = (uint64_t)M * N;
    const uint32_t total4 = (uint32_t)(slab / 4u); // (< 2^32: the launcher checks)
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < total4; e += gridDim.x * 256u) {
    const uint32_t col = e / m4s, m4 = e - col * m4s;
    const float4 *p = reinterpret_cast<const float4 *>(part + ((uint64_t)z * nsplit) * slab + (uint64_t)col * M) + m4;
    float4 s = p[0];
    for (uint32_t i = 1; i < nsplit; ++i) {
        const float4 q = p[(uint64_t)i * (slab / 4u)];
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    OUT *o = out + z * c_batch + (uint64_t)col * ldc + 4u * m4;
    struct alignas(2) h4 { _Float16 v[4]; }; // (an f16 result at any element-aligned address: gemm_f16_common.hpp half8_u)
    if (alpha != 1.f) { s.x *= alpha; s.y *= alpha; s.z *= alpha; s.w *= alpha; }
    if (beta != 0.f) { // out = alpha * sum + beta * out (wg_gemm_ex); beta == 0 never reads `out`
        float4 c;
        if constexpr (sizeof(OUT) == 4) c = wg_ld_u(reinterpret_cast<const float *>(o));
        else { const h4 t = *reinterpret_cast<const h4 *>(o); c = make_float4((float)t.v[0], (float)t.v[1], (float)t.v[2], (float)t.v[3]); }
        s.x = fmaf(beta, c.x, s.x); s.y = fmaf(beta, c.y, s.y); s.z = fmaf(beta, c.z, s.z); s.w = fmaf(beta, c.w, s.w);
    }
    if constexpr (sizeof(OUT) == 4) {
        wg_st_u(reinterpret_cast<float *>(o), s);
    } else {
        h4 r = { { (_Float16)s.x, (_Float16)s.y, (_Float16)s.z, (_Float16)s.w } };
        *reinterpret_cast<h4 *>(o) = r;
    }
    }
}

// out[r * rs + c * cs] = alpha * sum_s part[s][c][r]: reads run along r (contiguous), writes are rs apart -- the outputs this serves are small
__global__ __launch_bounds__(256) void splitk_reduce_strided_kernel(const float *__restrict__ part, uint32_t nsplit, uint32_t M, uint32_t N,
                                                                    float *__restrict__ out, uint32_t rs, uint32_t cs, uint64_t c_batch, float alpha) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= M) return;
    const uint32_t z = blockIdx.z;
    const uint64_t slab = (uint64_t)M * N;
    for (uint32_t col = blockIdx.y; col < N; col += gridDim.y) {
        const float *p = part + ((uint64_t)z * nsplit) * slab + (uint64_t)col * M + r;
        float s = p[0];
        for (uint32_t i = 1; i < nsplit; ++i) s += p[(uint64_t)i * slab];
        out[z * c_batch + (uint64_t)r * rs + (uint64_t)col * cs] = alpha != 1.f ? s * alpha : s;
    }
}

} // namespace

// How many K-splits to use (1 = none). `tiles` = output tiles x matrices, `slots` = workgroups the chip holds at once,
// `k_units` = K / (kernel's K granule), `min_units` = fewest granules worth a workgroup's prologue/epilogue.
uint32_t wg_splitk_plan(uint64_t tiles, uint32_t slots, uint32_t k_units, uint32_t min_units, uint64_t out_elems, uint64_t max_ws_bytes) {
    if (tiles == 0 || tiles * 2 > slots) return 1; // at least half the chip is busy already
    uint32_t s = (uint32_t)(slots / tiles);
    const uint32_t by_k = k_units / min_units;
    if (s > by_k) s = by_k;
    while (s > 1 && (uint64_t)s * out_elems * 4u > max_ws_bytes) --s;
    return s < 2 ? 1 : s;
}

int wg_splitk_reduce(wg_ctx *ctx, const float *part, uint32_t nsplit, uint32_t M, uint32_t N, uint32_t nmats, wg_dtype dtype, void *out,
                     uint32_t ldc, uint64_t c_batch, float alpha, float beta) {
    if (nmats > 65535) return wg_set_error(WG_ERR_UNSUPPORTED, "split-K reduce: more than 65535 matrices");
    if ((uint64_t)(M / 4u) * N >= (1ull << 32) - 65536ull * 256ull) return wg_set_error(WG_ERR_UNSUPPORTED, "split-K reduce: output of more than 2^34 elements");
    const uint64_t blocks = ((uint64_t)(M / 4u) * N + 255u) / 256u;
    const dim3 grid((uint32_t)(blocks < 65536u ? (blocks ? blocks : 1u) : 65536u), 1, nmats), block(256); // (larger outputs: grid-stride trips)
    if (dtype == WG_F32) hipLaunchKernelGGL(splitk_reduce_kernel<float>, grid, block, 0, ctx->stream, part, nsplit, M, N, (float *)out, ldc, c_batch, alpha, beta);
    else hipLaunchKernelGGL(splitk_reduce_kernel<_Float16>, grid, block, 0, ctx->stream, part, nsplit, M, N, (_Float16 *)out, ldc, c_batch, alpha, beta);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

int wg_splitk_reduce_strided(wg_ctx *ctx, const float *part, uint32_t nsplit, uint32_t M, uint32_t N, uint32_t nmats, float *out,
                             uint32_t row_stride, uint32_t col_stride, uint64_t c_batch, float alpha) {
    if (nmats > 65535) return wg_set_error(WG_ERR_UNSUPPORTED, "split-K reduce: more than 65535 matrices");
    const dim3 grid((M + 255u) / 256u, N < 65535u ? N : 65535u, nmats), block(256);
    hipLaunchKernelGGL(splitk_reduce_strided_kernel, grid, block, 0, ctx->stream, part, nsplit, M, N, out, row_stride, col_stride, c_batch, alpha);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}
