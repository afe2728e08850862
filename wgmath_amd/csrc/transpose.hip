// Out-of-place transpose of column-major matrices: dst (cols x rows) [c + r * ld_dst] = src (rows x cols) [r + c * ld_src].
// Used by the ROW_MAJOR operator surface (shape.wgsl:49-57) for the one case that is not a pure re-labelling of a
// column-major call (GemmTr on row-major operands needs op(B) = B^T, which the MFMA kernels do not stage): HBM-bound
// pre-pass, 2 * sizeof(T) * rows * cols bytes. 64 x 64 tiles through a padded LDS tile: reads walk rows of src (coalesced),
// writes walk rows of dst (coalesced).
#include "wg_internal.hpp"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T *__restrict__ src, uint32_t ld_src, uint64_t src_batch, T *__restrict__ dst,
                                                        uint32_t ld_dst, uint64_t dst_batch, uint32_t rows, uint32_t cols) {
    __shared__ T tile[64][65];
    const uint32_t r0 = blockIdx.x * 64u, c0 = blockIdx.y * 64u;
    const T *s = src + blockIdx.z * src_batch;
    T *d = dst + blockIdx.z * dst_batch;
    const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6; // 64 x 4
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t c = c0 + ty + 4u * i, r = r0 + tx;
        if (r < rows && c < cols) tile[ty + 4 * i][tx] = s[(uint64_t)c * ld_src + r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t r = r0 + ty + 4u * i, c = c0 + tx; // dst column index = src row r, dst row index = src column c
        if (r < rows && c < cols) d[(uint64_t)r * ld_dst + c] = tile[tx][ty + 4 * i];
    }
}

} // namespace

int wgk_transpose(wg_ctx *ctx, wg_dtype dtype, uint32_t rows, uint32_t cols, uint32_t nmats, const void *src, uint32_t ld_src,
                  uint64_t src_batch, void *dst, uint32_t ld_dst, uint64_t dst_batch) {
    if (rows == 0 || cols == 0 || nmats == 0) return WG_OK;
    const dim3 grid((rows + 63u) / 64u, (cols + 63u) / 64u, nmats), block(256);
    if (grid.y > 65535u || grid.z > 65535u) return wg_set_error(WG_ERR_UNSUPPORTED, "transpose: matrix too wide or too many matrices for one launch");
    if (dtype == WG_F32)
        hipLaunchKernelGGL(transpose_kernel<float>, grid, block, 0, ctx->stream, (const float *)src, ld_src, src_batch, (float *)dst, ld_dst,
                           dst_batch, rows, cols);
    else
        hipLaunchKernelGGL(transpose_kernel<_Float16>, grid, block, 0, ctx->stream, (const _Float16 *)src, ld_src, src_batch, (_Float16 *)dst,
                           ld_dst, dst_batch, rows, cols);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

// Copy of a column-major block with zero fill: dst (rd x cd, ld_dst) = src (rs x cs, ld_src) where it exists, 0 elsewhere, element by
// element (any alignment, any stride). The operator front-end (api.hip) stages views that are not vec4-aligned -- the ones
// GpuMatrix::slice / rows / column hand out for odd offsets and lengths (tensor.rs:574-626) -- into dense zero-padded copies with it and
// copies the result back: an HBM-bound pass over operands the kernels then read at full width.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void stage_copy_kernel(T *__restrict__ dst, uint32_t ld_dst, uint64_t dst_batch, uint32_t rd, uint32_t cd,
                                                         const T *__restrict__ src, uint32_t ld_src, uint64_t src_batch, uint32_t rs, uint32_t cs) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, z = blockIdx.z;
    if (i >= rd) return;
    for (uint32_t j = blockIdx.y; j < cd; j += gridDim.y)
        dst[z * dst_batch + (uint64_t)j * ld_dst + i] = (i < rs && j < cs) ? src[z * src_batch + (uint64_t)j * ld_src + i] : (T)0;
}
} // namespace

int wgk_stage_copy(wg_ctx *ctx, wg_dtype dtype, void *dst, uint32_t ld_dst, uint64_t dst_batch, uint32_t rd, uint32_t cd, const void *src,
                   uint32_t ld_src, uint64_t src_batch, uint32_t rs, uint32_t cs, uint32_t nmats) {
    if (rd == 0 || cd == 0 || nmats == 0) return WG_OK;
    if (nmats > 65535u) return wg_set_error(WG_ERR_UNSUPPORTED, "staging copy: more than 65535 matrices");
    const dim3 grid((rd + 255u) / 256u, cd < 65535u ? cd : 65535u, nmats), block(256);
    if (dtype == WG_F32)
        hipLaunchKernelGGL(stage_copy_kernel<float>, grid, block, 0, ctx->stream, (float *)dst, ld_dst, dst_batch, rd, cd, (const float *)src, ld_src, src_batch, rs, cs);
    else
        hipLaunchKernelGGL(stage_copy_kernel<_Float16>, grid, block, 0, ctx->stream, (_Float16 *)dst, ld_dst, dst_batch, rd, cd, (const _Float16 *)src, ld_src, src_batch,
                           rs, cs);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}
