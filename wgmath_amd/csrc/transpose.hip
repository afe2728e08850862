// Out-of-place transpose of column-major matrices: dst (cols x rows) [c + r * ld_dst] = src (rows x cols) [r + c * ld_src].
// Used by the ROW_MAJOR operator surface (shape.wgsl:49-57) for the one case that is not a pure re-labelling of a
// column-major call (GemmTr on row-major operands needs op(B) = B^T, which the MFMA kernels do not stage): HBM-bound
// pre-pass, 2 * sizeof(T) * rows * cols bytes. 64 x 64 tiles through a padded LDS tile: reads walk rows of src (coalesced),
// writes walk rows of dst (coalesced).
#include "wg_internal.hpp"
#include <type_traits>

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

// Copy of a column-major block with zero fill: dst (rd x cd, ld_dst) = src (rs x cs, ld_src) where it exists, 0 elsewhere -- any alignment, any
// stride, either side. The operator front-end (api.hip) stages operands whose LENGTHS the kernels do not take into zero-padded copies with it and copies
// padded results back, the f16 launcher pads operands the MFMA kernels do not take as they are (gemm_f16.hip), and wg_copy_view exposes it.
//
// Round 6: 16-byte accesses whatever the alignment (the element-by-element kernel ran at ~1.3 TB/s; the f16 one fell back to it whenever one side was only
// 8-byte aligned). global_load_dwordx4 / global_store_dwordx4 take any element-aligned address on this target (tools/cpp/unaligned_probe.hip,
// profiles/r06_unaligned_probe.txt: within 2 % of the aligned rate at 4-byte offsets, 0.88 of it at 2-byte offsets), so a thread moves 16 bytes of a column
// -- E = 4 f32 / 8 f16 elements counted from the column's first -- with one load and one store wherever the two sides lie. Only a chunk that is not whole on
// a side (the last rows of a length that is not a multiple of E; the edge of the zero padding) goes element by element there: a 16-byte access
// could run off the end of a buffer.
namespace {
typedef uint32_t wg_u32x4 __attribute__((ext_vector_type(4)));
// ES: bytes per element (2 or 4). CPB: columns per workgroup (1: 256 chunks of one column; 4: a wave per column, for short columns).
template <int ES, int CPB>
__global__ __launch_bounds__(256) void copy2d_kernel(char *__restrict__ dst, uint32_t ld_dst, uint64_t dst_batch, uint32_t rd, uint32_t cd,
                                                     const char *__restrict__ src, uint32_t ld_src, uint64_t src_batch, uint32_t rs, uint32_t cs) {
    constexpr uint32_t TX = 256u / CPB, E = 16u / ES;
    using elem_t = std::conditional_t<ES == 4, uint32_t, uint16_t>;
    using gvec = const __attribute__((address_space(1), aligned(ES))) wg_u32x4;
    using gvec_w = __attribute__((address_space(1), aligned(ES))) wg_u32x4;
    const uint32_t z = blockIdx.z;
    const uint64_t i0 = ((uint64_t)blockIdx.x * TX + (threadIdx.x % TX)) * E;
    if (i0 >= rd) return;
    for (uint32_t j = blockIdx.y * CPB + threadIdx.x / TX; j < cd; j += gridDim.y * CPB) {
        const uintptr_t D = (uintptr_t)dst + (z * dst_batch + (uint64_t)j * ld_dst + i0) * ES;
        const uintptr_t S = (uintptr_t)src + (z * src_batch + (uint64_t)j * ld_src + i0) * ES;
        const uint32_t have = j < cs && i0 < rs ? (uint32_t)min((uint64_t)E, rs - i0) : 0u; // source elements of this chunk
        wg_u32x4 v = { 0u, 0u, 0u, 0u };
        if (have == E) v = *reinterpret_cast<gvec *>(S);
        else if (have) {
            elem_t e[E];
#pragma unroll
            for (uint32_t k = 0; k < E; ++k) e[k] = k < have ? reinterpret_cast<const __attribute__((address_space(1))) elem_t *>(S)[k] : (elem_t)0;
            if constexpr (ES == 4) v = wg_u32x4{ e[0], e[1], e[2], e[3] };
            else v = wg_u32x4{ e[0] | (uint32_t)e[1] << 16, e[2] | (uint32_t)e[3] << 16, e[4] | (uint32_t)e[5] << 16, e[6] | (uint32_t)e[7] << 16 };
        }
        if (i0 + E <= rd) *reinterpret_cast<gvec_w *>(D) = v;
        else {
            const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (uint32_t k = 0; k < E; ++k)
                if (i0 + k < rd) reinterpret_cast<__attribute__((address_space(1))) elem_t *>(D)[k] = ES == 4 ? (elem_t)w[k] : (elem_t)(w[k >> 1] >> (16u * (k & 1u)));
        }
    }
}
} // namespace

int wgk_stage_copy(wg_ctx *ctx, wg_dtype dtype, void *dst, uint32_t ld_dst, uint64_t dst_batch, uint32_t rd, uint32_t cd, const void *src,
                   uint32_t ld_src, uint64_t src_batch, uint32_t rs, uint32_t cs, uint32_t nmats) {
    if (rd == 0 || cd == 0 || nmats == 0) return WG_OK;
    if (nmats > 65535u) return wg_set_error(WG_ERR_UNSUPPORTED, "staging copy: more than 65535 matrices");
    const uint32_t es = dtype == WG_F32 ? 4u : 2u;
    const uint64_t chunks = ((uint64_t)rd * es + 15u) / 16u;
    const bool narrow = chunks <= 64u;
    const uint32_t tx = narrow ? 64u : 256u, cpb = narrow ? 4u : 1u, gy = (cd + cpb - 1u) / cpb;
    const dim3 grid((uint32_t)((chunks + tx - 1u) / tx), gy < 65535u ? gy : 65535u, nmats), block(256);
    char *d = (char *)dst;
    const char *s = (const char *)src;
    if (es == 4u) {
        if (narrow) hipLaunchKernelGGL((copy2d_kernel<4, 4>), grid, block, 0, ctx->stream, d, ld_dst, dst_batch, rd, cd, s, ld_src, src_batch, rs, cs);
        else hipLaunchKernelGGL((copy2d_kernel<4, 1>), grid, block, 0, ctx->stream, d, ld_dst, dst_batch, rd, cd, s, ld_src, src_batch, rs, cs);
    } else {
        if (narrow) hipLaunchKernelGGL((copy2d_kernel<2, 4>), grid, block, 0, ctx->stream, d, ld_dst, dst_batch, rd, cd, s, ld_src, src_batch, rs, cs);
        else hipLaunchKernelGGL((copy2d_kernel<2, 1>), grid, block, 0, ctx->stream, d, ld_dst, dst_batch, rd, cd, s, ld_src, src_batch, rs, cs);
    }
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}
