// Batched application of the geometry header's functions (include/wgebra_geometry.hpp): one item per thread.
// This is the test / demo harness of a header whose real use is inside other kernels -- the reference tests its geometry
// shaders the same way (a one-invocation-per-matrix test kernel, e.g. cholesky.rs:60-72).
// Item layouts (dense f32, column-major matrices, no WGSL vec3 padding):
//   INV, CHOLESKY : in N*N                       -> out N*N
//   LU            : in N*N                       -> out N*N (lu) + N (ia) + N (ib) + 1 (len), indices as floats
//   QR            : in N*N                       -> out N*N (q) + N*N (r)
//   SYM_EIGEN     : in N*N (symmetric)           -> out N*N (eigenvectors) + N (eigenvalues)
//   SVD (N = 2,3) : in N*N                       -> out N*N (u) + N (s) + N*N (vt)
//   ROT2          : in 2 angles + vec2 (4)       -> out mul (2) + mulVec (2) + invMulVec (2) + toMatrix (4) + angle of mul (1) = 11
//   QUAT          : in 2 scaled axes + vec3 (9)  -> out mul (4) + mulVec (3) + invMulVec (3) + toMatrix (9) = 19
//   SIM2          : in 2 x (angle, t.x, t.y, scale) + pt (10) -> out mul (4: angle, t, scale) + inv (4) + mulPt (2) + invMulPt (2) + mulVec (2) = 14
//   SIM3          : in 2 x (axis(3), t(3), scale) + pt (17)   -> out mul (quat 4, t 3, scale 1) + inv (8) + mulPt (3) + invMulPt (3) + mulVec (3) = 25
// The transform functions one by one on RAW coordinates (what tests/golden/wgsl_exec_geometry.npz, executed from the reference's WGSL text, holds;
// layouts in geometry_items.hpp raw_item): QUAT_RAW 11 -> 27, ROT2_RAW 6 -> 12, SIM2_RAW 12 -> 18, SIM3_RAW 19 -> 28, FROM (scaled axis + angle) 4 -> 6,
// UTILS (trig + min_max) 19 -> 11, ROT2_EXT (angle, cancel_y, is_valid, rotate_rows3/4) 31 -> 29, EIGVALS2 4 -> 2; SVD_RECOMPOSE (dim 2, 3) 2 N*N + N -> N*N.
#include "wg_internal.hpp"

#include "geometry_items.hpp"

namespace {
static_assert(WG_GEOM_INV == wgg_items::OP_INV && WG_GEOM_SVD == wgg_items::OP_SVD && WG_GEOM_SIM3 == wgg_items::OP_SIM3 &&
                  WG_GEOM_QUAT_RAW == wgg_items::OP_QUAT_RAW && WG_GEOM_FROM == wgg_items::OP_FROM && WG_GEOM_UTILS == wgg_items::OP_UTILS &&
                  WG_GEOM_SVD_RECOMPOSE == wgg_items::OP_SVD_RECOMPOSE && WG_GEOM_SVD_RECOMPOSE == wgg_items::OP_LAST, "op numbering");

template <int N>
__global__ void geom_mat_kernel(int op, const float *__restrict__ in, float *__restrict__ out, uint32_t count, uint32_t in_stride, uint32_t out_stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) wgg_items::mat_item<N>(op, in + (uint64_t)i * in_stride, out + (uint64_t)i * out_stride);
}
__global__ void geom_transform_kernel(int op, const float *__restrict__ in, float *__restrict__ out, uint32_t count, uint32_t in_stride,
                                      uint32_t out_stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if (op >= wgg_items::OP_QUAT_RAW) wgg_items::raw_item(op, in + (uint64_t)i * in_stride, out + (uint64_t)i * out_stride);
    else wgg_items::transform_item(op, in + (uint64_t)i * in_stride, out + (uint64_t)i * out_stride);
}
} // namespace

extern "C" int wg_geometry_apply(wg_ctx *ctx, wg_geom_op op, uint32_t dim, const wg_buf *in, wg_buf *out, uint32_t count) {
    if (!ctx || !in || !out) return wg_set_error(WG_ERR_INVALID_ARG, "wg_geometry_apply: NULL argument");
    if ((int)op < 0 || (int)op > wgg_items::OP_LAST) return wg_set_error(WG_ERR_INVALID_ARG, "wg_geometry_apply: unknown op %d", (int)op);
    const bool is_mat = wgg_items::is_mat_op((int)op);
    if (is_mat && (dim < 2 || dim > 4)) return wg_set_error(WG_ERR_INVALID_ARG, "wg_geometry_apply: dim %u not in 2..4", dim);
    if ((op == WG_GEOM_SVD || op == WG_GEOM_SVD_RECOMPOSE) && dim == 4)
        return wg_set_error(WG_ERR_UNSUPPORTED, "wg_geometry_apply: the reference has svd2 and svd3 only");
    if (count == 0) return WG_OK;
    const uint32_t nin = wgg_items::in_floats(op, dim), nout = wgg_items::out_floats(op, dim);
    if (in->bytes < (size_t)count * nin * 4 || out->bytes < (size_t)count * nout * 4)
        return wg_set_error(WG_ERR_OUT_OF_BOUNDS, "wg_geometry_apply: %u items need %zu input and %zu output bytes", count,
                            (size_t)count * nin * 4, (size_t)count * nout * 4);
    WG_HIP_TRY(hipSetDevice(ctx->device));
    const dim3 grid((count + 255) / 256), block(256);
    const float *pi = (const float *)in->ptr;
    float *po = (float *)out->ptr;
    if (!is_mat) hipLaunchKernelGGL(geom_transform_kernel, grid, block, 0, ctx->stream, (int)op, pi, po, count, nin, nout);
    else if (dim == 2) hipLaunchKernelGGL(geom_mat_kernel<2>, grid, block, 0, ctx->stream, (int)op, pi, po, count, nin, nout);
    else if (dim == 3) hipLaunchKernelGGL(geom_mat_kernel<3>, grid, block, 0, ctx->stream, (int)op, pi, po, count, nin, nout);
    else hipLaunchKernelGGL(geom_mat_kernel<4>, grid, block, 0, ctx->stream, (int)op, pi, po, count, nin, nout);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}
