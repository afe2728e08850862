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

// Copy of a column-major block with zero fill: dst (rd x cd, ld_dst) = src (rs x cs, ld_src) where it exists, 0 elsewhere -- any alignment, any
// stride, either side. The operator front-end (api.hip) stages views that are not vec4-aligned -- the ones GpuMatrix::slice / rows / column
// hand out for odd offsets and lengths (tensor.rs:574-626) -- into dense zero-padded copies with it and copies the result back, and the f16
// launcher pads operands the MFMA kernels do not take as they are (gemm_f16.hip): HBM-bound passes over operands the kernels then read at full width.
//
// Round 6: 16-byte accesses whatever the alignment (the element-by-element kernel ran at ~1.3 TB/s; the f16 one fell back to it whenever
// one side was only 8-byte aligned). A thread owns one 16-byte-aligned chunk of a destination column; the source bytes that belong there start
// `m` bytes into an aligned 16-byte chunk of the source column (m is the same for the whole column), so it loads that chunk and the next one and
// shifts (v_alignbyte_b32). Aligned chunks that hold at least one byte of the source column are the only ones read: such a chunk lies in the
// same page as that byte. The ragged ends of a column (a first / last chunk that is only partly inside it) are masked and stored element by element.
namespace {
__device__ __forceinline__ uint4 shift_chunks(uint4 lo, uint4 hi, uint32_t m) { // bytes [m, m + 16) of lo:hi
    const uint32_t w[8] = { lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w };
    const uint32_t b = m & 3u;
    uint4 r;
    switch (m >> 2) { // (uniform over the wave)
    case 0: r = make_uint4(__builtin_amdgcn_alignbyte(w[1], w[0], b), __builtin_amdgcn_alignbyte(w[2], w[1], b), __builtin_amdgcn_alignbyte(w[3], w[2], b), __builtin_amdgcn_alignbyte(w[4], w[3], b)); break;
    case 1: r = make_uint4(__builtin_amdgcn_alignbyte(w[2], w[1], b), __builtin_amdgcn_alignbyte(w[3], w[2], b), __builtin_amdgcn_alignbyte(w[4], w[3], b), __builtin_amdgcn_alignbyte(w[5], w[4], b)); break;
    case 2: r = make_uint4(__builtin_amdgcn_alignbyte(w[3], w[2], b), __builtin_amdgcn_alignbyte(w[4], w[3], b), __builtin_amdgcn_alignbyte(w[5], w[4], b), __builtin_amdgcn_alignbyte(w[6], w[5], b)); break;
    default: r = make_uint4(__builtin_amdgcn_alignbyte(w[4], w[3], b), __builtin_amdgcn_alignbyte(w[5], w[4], b), __builtin_amdgcn_alignbyte(w[6], w[5], b), __builtin_amdgcn_alignbyte(w[7], w[6], b)); break;
    }
    return r;
}

// ES: bytes per element (2 or 4). CPB: columns per workgroup (1: 256 chunks of one column; 4: a wave per column, for short columns).
template <int ES, int CPB>
__global__ __launch_bounds__(256) void copy2d_kernel(char *__restrict__ dst, uint32_t ld_dst, uint64_t dst_batch, uint32_t rd, uint32_t cd,
                                                     const char *__restrict__ src, uint32_t ld_src, uint64_t src_batch, uint32_t rs, uint32_t cs) {
    constexpr uint32_t TX = 256u / CPB;
    const uint32_t z = blockIdx.z;
    const int64_t q = (int64_t)blockIdx.x * TX + (threadIdx.x % TX);
    const int64_t nb = (int64_t)rd * ES;
    for (uint32_t j = blockIdx.y * CPB + threadIdx.x / TX; j < cd; j += gridDim.y * CPB) {
        const uintptr_t D = (uintptr_t)dst + (z * dst_batch + (uint64_t)j * ld_dst) * ES;
        const int64_t o = 16 * q - (int64_t)(D & 15u); // this chunk's first byte, as an offset into the column (< 0: the chunk starts ahead of it)
        if (o >= nb) continue;
        const uintptr_t S = (uintptr_t)src + (z * src_batch + (uint64_t)j * ld_src) * ES;
        const int64_t vb = j < cs ? (int64_t)rs * ES : 0; // bytes of the source column
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (o + 16 > 0 && o < vb) {
            const uintptr_t sa = S + o;
            const uint32_t m = (uint32_t)(sa & 15u);
            const int64_t ol = o - (int64_t)m; // column offset of the aligned source chunk `lo`
            uint4 lo = make_uint4(0u, 0u, 0u, 0u), hi = lo;
            if (ol + 16 > 0 && ol < vb) lo = *reinterpret_cast<const uint4 *>(sa - m);
            if (m && ol + 32 > 0 && ol + 16 < vb) hi = *reinterpret_cast<const uint4 *>(sa - m + 16);
            v = shift_chunks(lo, hi, m);
            if (o < 0 || o + 16 > vb) { // an end of the source column: what lies outside it is zero, whatever the chunks held
                uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                for (int e = 0; e < 16 / ES; ++e) {
                    const int64_t at = o + e * ES;
                    if (at < 0 || at >= vb) w[(e * ES) >> 2] &= ES == 4 ? 0u : (e & 1 ? 0x0000ffffu : 0xffff0000u);
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        if (o >= 0 && o + 16 <= nb) *reinterpret_cast<uint4 *>(D + o) = v;
        else {
            const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int e = 0; e < 16 / ES; ++e) {
                const int64_t at = o + e * ES;
                if (at < 0 || at >= nb) continue;
                if (ES == 4) *reinterpret_cast<uint32_t *>(D + at) = w[e];
                else *reinterpret_cast<uint16_t *>(D + at) = (uint16_t)(w[e >> 1] >> (16 * (e & 1)));
            }
        }
    }
}
} // namespace

int wgk_stage_copy(wg_ctx *ctx, wg_dtype dtype, void *dst, uint32_t ld_dst, uint64_t dst_batch, uint32_t rd, uint32_t cd, const void *src,
                   uint32_t ld_src, uint64_t src_batch, uint32_t rs, uint32_t cs, uint32_t nmats) {
    if (rd == 0 || cd == 0 || nmats == 0) return WG_OK;
    if (nmats > 65535u) return wg_set_error(WG_ERR_UNSUPPORTED, "staging copy: more than 65535 matrices");
    const uint32_t es = dtype == WG_F32 ? 4u : 2u;
    const uint64_t chunks = ((uint64_t)rd * es + 15u) / 16u + 1u; // (+ 1: a column that starts inside a chunk ends in one more)
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
