// Operator front-end of the C ABI: the host-side half of Gemm/Gemv/Reduce/OpAssign::dispatch.
// Mirrors, check for check, wgebra gemm.rs:75-126, gemv.rs:74-136, reduce.rs:100-113, op_assign.rs:79-94 and the
// silent-skip rules of wgcore kernel.rs:111-123,144; adds the bounds/alignment checks the reference leaves to UB.
#include "wg_internal.hpp"

namespace {

struct View {
    uint32_t rows, cols, mats, stride, stride_mat, offset;
};
inline View mk(const wg_view_shape &s) { return { s.size[0], s.size[1], s.size[2], s.stride, s.stride_mat, s.offset }; }

// Largest element index the view touches + 1 (0 for an empty view).
inline uint64_t extent(const View &v) {
    if (v.rows == 0 || v.cols == 0 || v.mats == 0) return 0;
    return (uint64_t)(v.mats - 1) * v.stride_mat + v.offset + (uint64_t)(v.rows - 1) + (uint64_t)(v.cols - 1) * v.stride + 1;
}

int check_bounds(const char *op, const char *name, const View &v, const wg_buf *b, wg_dtype dt) {
    const uint64_t need = extent(v), have = b->bytes / wg_dtype_size(dt);
    if (need > have)
        return wg_set_error(WG_ERR_OUT_OF_BOUNDS, "%s: view `%s` addresses %llu elements but its buffer holds %llu", op, name,
                            (unsigned long long)need, (unsigned long long)have);
    return WG_OK;
}

// The reference's kernels bind every buffer as array<vec4<f32>> (gemm.wgsl:9-14, gemv.wgsl:9-14) and convert shapes with
// with_vec4_elts (shape.wgsl:64-66): rows, stride, stride_mat and offset must be multiples of 4 for that to address
// the same elements. (stride only matters with > 1 column, stride_mat with > 1 matrix: GpuMatrix::column sets
// stride = 1 and GpuCubeView::matrix sets stride_mat = 1, tensor.rs:474,574-584.) Views like that go to the kernels as they
// are (16-byte accesses); all others are staged (gemm_staged / gemv_staged below).
inline bool vec4_ok(const View &v) { return !(v.rows % 4 || v.offset % 4 || (v.cols > 1 && v.stride % 4) || (v.mats > 1 && v.stride_mat % 4)); }
int check_common(const char *op, const wg_ctx *ctx, wg_dtype dtype, const wg_buf *const *bufs, int n) {
    if (!ctx) return wg_set_error(WG_ERR_INVALID_ARG, "%s: ctx is NULL", op);
    if (dtype != WG_F32 && dtype != WG_F16) return wg_set_error(WG_ERR_INVALID_ARG, "%s: unknown dtype %d", op, (int)dtype);
    for (int i = 0; i < n; ++i) {
        if (!bufs[i]) return wg_set_error(WG_ERR_INVALID_ARG, "%s: buffer argument %d is NULL", op, i);
        if (bufs[i]->ctx->device != ctx->device)
            return wg_set_error(WG_ERR_INVALID_ARG, "%s: buffer argument %d lives on device %d, the context on device %d", op, i,
                                bufs[i]->ctx->device, ctx->device);
    }
    return WG_OK;
}

inline const void *elem_ptr(const wg_buf *b, uint64_t elem, wg_dtype dt) { return (const char *)b->ptr + elem * wg_dtype_size(dt); }

inline uint32_t up8(uint32_t x) { return (x + 7u) & ~7u; }

// Views that are not vec4-aligned. The reference's kernels cannot address them (they bind array<vec4<f32>> and divide rows, strides
// and offsets by 4, shape.wgsl:64-66: a slice at an odd row, a column block with an odd stride or a length that is not a multiple of
// 4 reads the wrong elements there), yet its own constructors hand them out (GpuMatrix::slice / rows / column, tensor.rs:574-626).
// Here they compute op(A) B exactly as the aligned call would. Offsets, leading dimensions and batch strides are free (the kernels' 16-byte
// accesses and LDS-DMA take element-aligned addresses); an operand that carries a LENGTH that is not a multiple of 4 is staged into a dense
// zero-padded copy (the length rounded up to 8; zeros add nothing to a dot product), the usual kernels run, and a padded result is copied back
// into the output view -- HBM-bound passes in a fourth context scratch (it cannot grow inside a recording).
int gemm_staged(wg_ctx *ctx, bool tr, wg_dtype dtype, float alpha, float beta, wg_buf *out, const View &o, const wg_buf *m1, const View &a, const wg_buf *m2,
                const View &b, uint32_t M, uint32_t N, uint32_t K) {
    const size_t es = wg_dtype_size(dtype);
    // Only the operands that need it are copied (round 6: a 16384 x 16384 matrix times ONE column -- N % 4 != 0, nothing else -- paid 800 us for the copy
    // of the matrix, 177 us now): a dimension that is not a multiple of 4 is rounded up to 8 in the two operands that carry it, an operand
    // whose own view is not vec4-aligned is copied at the (possibly padded) sizes, the others are used where they lie.
    const uint32_t Mp = M % 4 ? up8(M) : M, Np = N % 4 && !(dtype == WG_F16 || M > 128u) ? up8(N) : N, Kp = K % 4 ? up8(K) : K, mats = o.mats; // (f16: any N as it is)
    // (the kernels take any offset / leading dimension / batch stride -- element-aligned LDS-DMA and 16-byte accesses --: only lengths are padded)
    const bool sa = Mp != M || Kp != K, sb = Kp != K || Np != N, sc = Mp != M || Np != N;
    const uint64_t ae = sa ? (uint64_t)Mp * Kp : 0, be = sb ? (uint64_t)Kp * Np : 0, ce = sc ? (uint64_t)Mp * Np : 0;
    if (ae * mats >= (1ull << 32) || be * mats >= (1ull << 32) || ce * mats >= (1ull << 32))
        return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: operands too large for the staging path of views that are not vec4-aligned");
    void *ws = nullptr;
    if (int rc = wg_ctx_stage_workspace(ctx, (size_t)((ae + be + ce) * mats * es), &ws)) return rc;
    char *ap = (char *)ws, *bp = ap + ae * mats * es, *cp = bp + be * mats * es;
    const uint32_t a_ld = tr ? Kp : Mp;
    wgk_mat A = { elem_ptr(m1, a.offset, dtype), a.stride, a.stride_mat }, B = { elem_ptr(m2, b.offset, dtype), b.stride, b.stride_mat };
    void *C = (void *)elem_ptr(out, o.offset, dtype);
    uint32_t ldc = o.stride;
    uint64_t c_batch = o.stride_mat;
    if (sa) {
        if (int rc = wgk_stage_copy(ctx, dtype, ap, a_ld, ae, tr ? Kp : Mp, tr ? Mp : Kp, A.ptr, a.stride, a.stride_mat, a.rows, a.cols, mats)) return rc;
        A = wgk_mat{ ap, a_ld, ae };
    }
    if (sb) {
        if (int rc = wgk_stage_copy(ctx, dtype, bp, Kp, be, Kp, Np, B.ptr, b.stride, b.stride_mat, b.rows, b.cols, mats)) return rc;
        B = wgk_mat{ bp, Kp, be };
    }
    if (sc) {
        if (beta != 0.f)
            if (int rc = wgk_stage_copy(ctx, dtype, cp, Mp, ce, Mp, Np, C, o.stride, o.stride_mat, M, N, mats)) return rc;
        C = cp; ldc = Mp; c_batch = ce;
    }
    if (int rc = dtype == WG_F32 ? wgk_gemm_f32(ctx, tr, Mp, Np, Kp, mats, (float *)C, ldc, c_batch, A, B, alpha, beta)
                                 : wgk_gemm_f16(ctx, tr, Mp, Np, Kp, mats, (__half *)C, ldc, c_batch, A, B, alpha, beta))
        return rc;
    if (!sc) return WG_OK;
    return wgk_stage_copy(ctx, dtype, (void *)elem_ptr(out, o.offset, dtype), o.stride, o.stride_mat, M, N, cp, Mp, ce, Mp, Np, mats);
}

int gemv_staged(wg_ctx *ctx, bool tr, wg_dtype dtype, wg_buf *out, const View &o, const wg_buf *m, const View &mm, const wg_buf *v, const View &vv,
                uint32_t rows_out, uint32_t k) {
    const size_t es = wg_dtype_size(dtype);
    // (as gemm_staged: only what needs it is copied -- a vector at an odd offset no longer costs a copy of the matrix)
    const uint32_t Op = rows_out % 4 ? up8(rows_out) : rows_out, Kp = k % 4 ? up8(k) : k, nrhs = o.cols, mats = o.mats;
    const uint32_t Rp = tr ? Kp : Op, Cp = tr ? Op : Kp;
    const bool sm = !vec4_ok(mm) || Op != rows_out || Kp != k, sv = !vec4_ok(vv) || Kp != k, so = !vec4_ok(o) || Op != rows_out;
    const uint64_t me = sm ? (uint64_t)Rp * Cp : 0, ve = sv ? (uint64_t)Kp * nrhs : 0, oe = so ? (uint64_t)Op * nrhs : 0;
    if (me * mats >= (1ull << 32) || ve * mats >= (1ull << 32) || oe * mats >= (1ull << 32))
        return wg_set_error(WG_ERR_UNSUPPORTED, "Gemv: operands too large for the staging path of views that are not vec4-aligned");
    // the matrix itself is off (offset, leading dimension or lengths): one pass over it where it lies, vectors and result addressed element by element (gemv_any.hip)
    if (sm) {
        if (k == 0) return wgk_stage_copy(ctx, dtype, (void *)elem_ptr(out, o.offset, dtype), o.stride, o.stride_mat, rows_out, nrhs, out->ptr, 1, 0, 0, 0, mats); // (an empty sum)
        return wgk_gemv_any(ctx, tr, dtype, mm.rows, mm.cols, nrhs, mats, (void *)elem_ptr(out, o.offset, dtype), o.stride, o.stride_mat,
                            wgk_mat{ elem_ptr(m, mm.offset, dtype), mm.stride, mm.stride_mat }, wgk_mat{ elem_ptr(v, vv.offset, dtype), vv.stride, vv.stride_mat });
    }
    void *ws = nullptr;
    if (int rc = wg_ctx_stage_workspace(ctx, (size_t)((me + ve + oe) * mats * es), &ws)) return rc;
    char *mp = (char *)ws, *vp = mp + me * mats * es, *op = vp + ve * mats * es;
    wgk_mat Mx = { elem_ptr(m, mm.offset, dtype), mm.stride, mm.stride_mat }, Vx = { elem_ptr(v, vv.offset, dtype), vv.stride, vv.stride_mat };
    void *O = (void *)elem_ptr(out, o.offset, dtype);
    uint32_t ldo = o.stride;
    uint64_t o_batch = o.stride_mat;
    if (sm) {
        if (int rc = wgk_stage_copy(ctx, dtype, mp, Rp, me, Rp, Cp, Mx.ptr, mm.stride, mm.stride_mat, mm.rows, mm.cols, mats)) return rc;
        Mx = wgk_mat{ mp, Rp, me };
    }
    if (sv) {
        if (int rc = wgk_stage_copy(ctx, dtype, vp, Kp, ve, Kp, nrhs, Vx.ptr, vv.stride, vv.stride_mat, k, nrhs, mats)) return rc;
        Vx = wgk_mat{ vp, Kp, ve };
    }
    if (so) { O = op; ldo = Op; o_batch = oe; }
    if (int rc = wgk_gemv(ctx, tr, dtype, Op, Kp, nrhs, mats, O, ldo, o_batch, Mx, Vx)) return rc;
    if (!so) return WG_OK;
    return wgk_stage_copy(ctx, dtype, (void *)elem_ptr(out, o.offset, dtype), o.stride, o.stride_mat, rows_out, nrhs, op, Op, oe, Op, nrhs, mats);
}

} // namespace

int wg_gemm_f16_panels(wg_ctx *ctx, bool tr, void *out_panel0, uint32_t ldc, const wg_buf *m1, wg_view_shape m1_shape, const wg_buf *m2, wg_view_shape m2_shape,
                       const wgk_panels &panels) {
    const View a = mk(m1_shape), b = mk(m2_shape);
    const uint32_t m_rows = tr ? a.cols : a.rows, m_cols = tr ? a.rows : a.cols;
    if (m_cols != b.rows || a.mats != 1 || b.mats != 1 || m1->bytes == 0 || m2->bytes == 0) return WG_ERR_UNSUPPORTED;
    if (!vec4_ok(a) || !vec4_ok(b) || m_rows % 4 || m_cols % 4 || b.cols % 4) return WG_ERR_UNSUPPORTED;
    if (int rc = check_bounds("Gemm", "m1", a, m1, WG_F16)) return rc;
    if (int rc = check_bounds("Gemm", "m2", b, m2, WG_F16)) return rc;
    const wgk_mat A = { elem_ptr(m1, a.offset, WG_F16), a.stride, a.stride_mat }, B = { elem_ptr(m2, b.offset, WG_F16), b.stride, b.stride_mat };
    return wgk_gemm_f16(ctx, tr, m_rows, b.cols, m_cols, 1, (__half *)out_panel0, ldc, 0, A, B, 1.f, 0.f, &panels);
}

extern "C" {

int wg_gemm(wg_ctx *ctx, wg_gemm_variant variant, wg_dtype dtype, wg_buf *out, wg_view_shape out_shape, const wg_buf *m1,
            wg_view_shape m1_shape, const wg_buf *m2, wg_view_shape m2_shape) {
    return wg_gemm_ex(ctx, variant, dtype, 1.f, 0.f, out, out_shape, m1, m1_shape, m2, m2_shape);
}

int wg_gemm_ex(wg_ctx *ctx, wg_gemm_variant variant, wg_dtype dtype, float alpha, float beta, wg_buf *out, wg_view_shape out_shape,
               const wg_buf *m1, wg_view_shape m1_shape, const wg_buf *m2, wg_view_shape m2_shape) {
    const wg_buf *bufs[3] = { out, m1, m2 };
    if (int rc = check_common("Gemm", ctx, dtype, bufs, 3)) return rc;
    if ((int)variant < 0 || (int)variant > 3) return wg_set_error(WG_ERR_INVALID_ARG, "Gemm: unknown variant %d", (int)variant);
    const bool tr = variant == WG_GEMM_TR || variant == WG_GEMM_TR_FAST;
    const View o = mk(out_shape), a = mk(m1_shape), b = mk(m2_shape);

    // gemm.rs:81-96
    const uint32_t m_rows = tr ? a.cols : a.rows, m_cols = tr ? a.rows : a.cols;
    if (m_cols != b.rows || m_rows != o.rows || o.cols != b.cols || o.mats != a.mats || o.mats != b.mats)
        return wg_set_error(WG_ERR_DIM_MISMATCH,
                            "Gemm: dimension mismatch. (out [%u,%u,%u], m1 [%u,%u,%u]%s, m2 [%u,%u,%u])", o.rows, o.cols, o.mats,
                            a.rows, a.cols, a.mats, tr ? "^T" : "", b.rows, b.cols, b.mats);
    // kernel.rs:111-123 (zero-sized binding) and :144 (zero-sized grid): silently skipped
    if (out->bytes == 0 || m1->bytes == 0 || m2->bytes == 0) return WG_OK;
    if (o.rows == 0 || o.mats == 0) return WG_OK;
    if (o.cols == 0) return WG_OK; // the kernels' k-loop over m2 columns runs zero times

    if (int rc = check_bounds("Gemm", "out", o, out, dtype)) return rc;
    if (int rc = check_bounds("Gemm", "m1", a, m1, dtype)) return rc;
    if (int rc = check_bounds("Gemm", "m2", b, m2, dtype)) return rc;

    WG_HIP_TRY(hipSetDevice(ctx->device));
    // lengths the kernels' 4 x 4 blocks do not take (gemm.wgsl:87,94): zero-padded copies of the operands that carry them
    // (1 .. 7 columns that are not a multiple of 4 on otherwise aligned views: exactly a Gemv with that many right-hand sides -- no copy of anything)
    if (vec4_ok(o) && vec4_ok(a) && vec4_ok(b) && m_cols % 4 == 0 && m_rows % 4 == 0 && o.cols % 4 && o.cols < 8 && alpha == 1.f && beta == 0.f) // (the tuned Gemv kernels: aligned views)
        return wgk_gemv(ctx, tr, dtype, m_rows, m_cols, o.cols, o.mats, (void *)elem_ptr(out, o.offset, dtype), o.stride, o.stride_mat,
                        wgk_mat{ elem_ptr(m1, a.offset, dtype), a.stride, a.stride_mat }, wgk_mat{ elem_ptr(m2, b.offset, dtype), b.stride, b.stride_mat });
    // (the kernels take any offset, leading dimension and batch stride since round 6 -- gemm_f16.hip, gemm_f32*.hip: LDS-DMA and 16-byte accesses at element-aligned
    // addresses -- so only the lengths count)
    // (f16: the kernels take any number of columns -- they clamp their loads of m2 per column and skip the stores past N --, so N is as free as the offsets;
    //  f32: likewise from 129 rows on; the few-row forms of its launcher compute the transposed product and need N % 4 == 0 --
    //  tests/test_gpu_parity.py::test_gemm_f32_any_number_of_columns fails on them without the copies)
    const bool n_free = dtype == WG_F16 || m_rows > 128u; // (f32 with up to 128 rows: the few-row forms compute the transposed product, where N is the length that must be a multiple of 4)
    if ((o.cols % 4 && !n_free) || m_cols % 4 || m_rows % 4)
        return gemm_staged(ctx, tr, dtype, alpha, beta, out, o, m1, a, m2, b, m_rows, o.cols, m_cols);
    wgk_mat A = { elem_ptr(m1, a.offset, dtype), a.stride, a.stride_mat };
    wgk_mat B = { elem_ptr(m2, b.offset, dtype), b.stride, b.stride_mat };
    void *C = (void *)elem_ptr(out, o.offset, dtype);
    if (dtype == WG_F32) return wgk_gemm_f32(ctx, tr, m_rows, o.cols, m_cols, o.mats, (float *)C, o.stride, o.stride_mat, A, B, alpha, beta);
    return wgk_gemm_f16(ctx, tr, m_rows, o.cols, m_cols, o.mats, (__half *)C, o.stride, o.stride_mat, A, B, alpha, beta);
}

int wg_gemv(wg_ctx *ctx, wg_gemv_variant variant, wg_dtype dtype, wg_buf *out, wg_view_shape out_shape, const wg_buf *m,
            wg_view_shape m_shape, const wg_buf *v, wg_view_shape v_shape) {
    const wg_buf *bufs[3] = { out, m, v };
    if (int rc = check_common("Gemv", ctx, dtype, bufs, 3)) return rc;
    if ((int)variant < 0 || (int)variant > 3) return wg_set_error(WG_ERR_INVALID_ARG, "Gemv: unknown variant %d", (int)variant);
    const bool tr = variant == WG_GEMV_TR || variant == WG_GEMV_TR_FAST;
    const View o = mk(out_shape), mm = mk(m_shape), vv = mk(v_shape);

    // gemv.rs:79-91 -- the reference checks exactly these two
    const uint32_t m_rows = tr ? mm.cols : mm.rows, m_cols = tr ? mm.rows : mm.cols;
    if (m_cols != vv.rows || m_rows != o.rows)
        return wg_set_error(WG_ERR_DIM_MISMATCH, "Gemv: dimension mismatch. (out [%u,%u,%u], m [%u,%u,%u]%s, v [%u,%u,%u])", o.rows,
                            o.cols, o.mats, mm.rows, mm.cols, mm.mats, tr ? "^T" : "", vv.rows, vv.cols, vv.mats);
    // gemv.rs:99-104: GemvTrFast silently becomes GemvTr when m.rows % 128 != 0 (same kernel here either way)
    if (variant == WG_GEMV_TR_FAST && mm.rows % 128u != 0) variant = WG_GEMV_TR;
    // gemv.rs:122: assert_eq!(out_nrows % 4, 0) on the fast paths
    if ((variant == WG_GEMV_FAST || variant == WG_GEMV_TR_FAST) && o.rows % 4u != 0)
        return wg_set_error(WG_ERR_PRECONDITION, "Gemv: assertion `left == right` failed (out_nrows %% 4 == 0, gemv.rs:122): out has %u rows",
                            o.rows);
    if (out->bytes == 0 || m->bytes == 0 || v->bytes == 0) return WG_OK;
    if (o.rows == 0 || o.cols == 0 || o.mats == 0) return WG_OK;

    // The grid is [.., out_ncols, out_nmats] and indexes m with z, v with (y, z) (gemv.wgsl:40,46,62): the views
    // that are actually addressed take their column / matrix counts from `out`.
    const View m_eff = { mm.rows, mm.cols, o.mats, mm.stride, mm.stride_mat, mm.offset };
    const View v_eff = { vv.rows, o.cols, o.mats, vv.stride, vv.stride_mat, vv.offset };
    if (int rc = check_bounds("Gemv", "out", o, out, dtype)) return rc;
    if (int rc = check_bounds("Gemv", "m", m_eff, m, dtype)) return rc;
    if (int rc = check_bounds("Gemv", "v", v_eff, v, dtype)) return rc;

    WG_HIP_TRY(hipSetDevice(ctx->device));
    // views / sizes the vec4 kernels cannot address as they are (shape.wgsl:64-66; gemv.wgsl:73,76): the any-alignment kernels (a matrix that is off), or copies of the vectors
    if (!vec4_ok(o) || !vec4_ok(m_eff) || !vec4_ok(v_eff) || m_cols % 4 || m_rows % 4)
        return gemv_staged(ctx, tr, dtype, out, o, m, m_eff, v, v_eff, m_rows, m_cols);
    wgk_mat M = { elem_ptr(m, mm.offset, dtype), mm.stride, mm.stride_mat };
    wgk_mat V = { elem_ptr(v, vv.offset, dtype), vv.stride, vv.stride_mat };
    return wgk_gemv(ctx, tr, dtype, o.rows, m_cols, o.cols, o.mats, (void *)elem_ptr(out, o.offset, dtype), o.stride, o.stride_mat, M, V);
}

int wg_reduce(wg_ctx *ctx, wg_reduce_op op, wg_dtype dtype, const wg_buf *value, wg_view_shape value_shape, wg_buf *result) {
    const wg_buf *bufs[2] = { value, result };
    if (int rc = check_common("Reduce", ctx, dtype, bufs, 2)) return rc;
    if ((int)op < 0 || (int)op > 4) return wg_set_error(WG_ERR_INVALID_ARG, "Reduce: unknown op %d", (int)op);
    if (value->bytes == 0 || result->bytes == 0) return WG_OK; // kernel.rs:111-123
    if (result->bytes < wg_dtype_size(dtype)) return wg_set_error(WG_ERR_OUT_OF_BOUNDS, "Reduce: result buffer smaller than one element");
    // reduce.wgsl:71-72: input[offset + i], i < nrows; stride / ncols / nmats are ignored
    const View vec = { value_shape.size[0], 1, 1, 1, 1, value_shape.offset };
    if (int rc = check_bounds("Reduce", "value", vec, value, dtype)) return rc;
    WG_HIP_TRY(hipSetDevice(ctx->device));
    return wgk_reduce(ctx, (int)op, dtype, elem_ptr(value, vec.offset, dtype), vec.rows, 1, 1, 0, 0, result->ptr);
}

int wg_reduce_fast(wg_ctx *ctx, wg_reduce_op op, wg_dtype dtype, const wg_buf *value, wg_view_shape value_shape, wg_buf *result) {
    const wg_buf *bufs[2] = { value, result };
    if (int rc = check_common("Reduce", ctx, dtype, bufs, 2)) return rc;
    if ((int)op < 0 || (int)op > 4) return wg_set_error(WG_ERR_INVALID_ARG, "Reduce: unknown op %d", (int)op);
    if (value->bytes == 0 || result->bytes == 0) return WG_OK; // kernel.rs:111-123
    if (result->bytes < wg_dtype_size(dtype)) return wg_set_error(WG_ERR_OUT_OF_BOUNDS, "Reduce: result buffer smaller than one element");
    const View vec = { value_shape.size[0], 1, 1, 1, 1, value_shape.offset };
    if (int rc = check_bounds("Reduce", "value", vec, value, dtype)) return rc;
    WG_HIP_TRY(hipSetDevice(ctx->device));
    return wgk_reduce_fast(ctx, (int)op, dtype, elem_ptr(value, vec.offset, dtype), vec.rows, result->ptr);
}

int wg_gemv_reduce(wg_ctx *ctx, wg_gemv_variant variant, wg_reduce_op op, wg_dtype dtype, wg_buf *result, const wg_buf *m,
                   wg_view_shape m_shape, const wg_buf *v, wg_view_shape v_shape) {
    // result[0] = reduce(op, op(m) v): Gemv into a context-owned scratch vector, then Reduce in the reference order on the same
    // stream -- one call, no intermediate tensor for the caller; bit-identical to wg_gemv followed by wg_reduce.
    const wg_buf *bufs[3] = { result, m, v };
    if (int rc = check_common("Gemv", ctx, dtype, bufs, 3)) return rc;
    if ((int)op < 0 || (int)op > 4) return wg_set_error(WG_ERR_INVALID_ARG, "Reduce: unknown op %d", (int)op);
    if ((int)variant < 0 || (int)variant > 3) return wg_set_error(WG_ERR_INVALID_ARG, "Gemv: unknown variant %d", (int)variant);
    if (result->bytes < wg_dtype_size(dtype)) return wg_set_error(WG_ERR_OUT_OF_BOUNDS, "Reduce: result buffer smaller than one element");
    const bool tr = variant == WG_GEMV_TR || variant == WG_GEMV_TR_FAST;
    const View mm = mk(m_shape), vv = mk(v_shape);
    if (vv.cols != 1 || vv.mats != 1 || mm.mats != 1)
        return wg_set_error(WG_ERR_UNSUPPORTED, "gemv_reduce: one matrix and one vector only");
    const uint32_t out_rows = tr ? mm.cols : mm.rows;
    void *ws = nullptr;
    if (int rc = wg_ctx_tr_workspace(ctx, (size_t)(out_rows ? out_rows : 4) * sizeof(float), &ws)) return rc;
    // Launch-bound sizes (the single-kernel Gemv family): ONE launch -- the last workgroup to finish folds y in the reference order
    // (gemv.hip, gemv_n_small_reduce_kernel). Same checks as wg_gemv would make; everything else falls through to the two launches.
    if (!tr && dtype == WG_F32 && mm.cols == vv.rows && out_rows >= 128u && out_rows % 4u == 0 && mm.cols % 4u == 0 && mm.cols > 0 &&
        m->bytes && v->bytes && mm.offset % 4u == 0 && vv.offset % 4u == 0 && (mm.cols == 1 || mm.stride % 4u == 0) &&
        (uint64_t)out_rows * mm.cols <= (4ull << 20)) {
        const View m_eff = { mm.rows, mm.cols, 1, mm.stride, mm.stride_mat, mm.offset }, v_eff = { vv.rows, 1, 1, vv.stride, vv.stride_mat, vv.offset };
        if (int rc = check_bounds("Gemv", "m", m_eff, m, dtype)) return rc;
        if (int rc = check_bounds("Gemv", "v", v_eff, v, dtype)) return rc;
        if (!ctx->flags) {
            if (ctx->recording) return wg_set_error(WG_ERR_WORKSPACE, "gemv_reduce: arrival counter needed while recording: run the call once outside the recording first");
            WG_HIP_TRY(hipSetDevice(ctx->device));
            WG_HIP_TRY(hipMalloc((void **)&ctx->flags, 256));
            WG_HIP_TRY(hipMemsetAsync(ctx->flags, 0, 256, ctx->stream));
        }
        wgk_mat M = { elem_ptr(m, mm.offset, dtype), mm.stride, mm.stride_mat }, V = { elem_ptr(v, vv.offset, dtype), vv.stride, vv.stride_mat };
        const int rc = wgk_gemv_small_reduce(ctx, (int)op, out_rows, mm.cols, (float *)ws, M, V, ctx->flags, (float *)result->ptr);
        if (rc != WG_ERR_UNSUPPORTED) return rc;
    }
    wg_buf tmp;
    tmp.ctx = ctx; tmp.ptr = ws; tmp.bytes = (size_t)(out_rows ? out_rows : 4) * sizeof(float); tmp.usage = 0; tmp.owned = false; tmp.host_pinned = false;
    wg_view_shape os;
    os.size[0] = out_rows; os.size[1] = 1; os.size[2] = 1; os.stride = out_rows; os.stride_mat = out_rows; os.offset = 0;
    if (int rc = wg_gemv(ctx, variant, dtype, &tmp, os, m, m_shape, v, v_shape)) return rc;
    return wg_reduce(ctx, op, dtype, &tmp, os, result);
}

int wg_reduce_batched(wg_ctx *ctx, wg_reduce_op op, wg_dtype dtype, const wg_buf *values, wg_view_shape values_shape, wg_buf *results) {
    const wg_buf *bufs[2] = { values, results };
    if (int rc = check_common("Reduce", ctx, dtype, bufs, 2)) return rc;
    if ((int)op < 0 || (int)op > 4) return wg_set_error(WG_ERR_INVALID_ARG, "Reduce: unknown op %d", (int)op);
    const View v = mk(values_shape);
    const uint64_t nvec = (uint64_t)v.cols * v.mats;
    if (nvec == 0 || results->bytes == 0) return WG_OK;
    if (results->bytes / wg_dtype_size(dtype) < nvec)
        return wg_set_error(WG_ERR_OUT_OF_BOUNDS, "Reduce: results buffer holds %zu elements but %llu vectors are reduced",
                            results->bytes / wg_dtype_size(dtype), (unsigned long long)nvec);
    if (values->bytes == 0 && v.rows != 0) return WG_OK;
    if (int rc = check_bounds("Reduce", "values", v, values, dtype)) return rc;
    WG_HIP_TRY(hipSetDevice(ctx->device));
    return wgk_reduce(ctx, (int)op, dtype, elem_ptr(values, v.offset, dtype), v.rows, v.cols, v.mats, v.stride, v.stride_mat,
                      results->ptr);
}

int wg_op_assign(wg_ctx *ctx, wg_op_assign_variant op, wg_dtype dtype, wg_buf *a, wg_view_shape a_shape, const wg_buf *b,
                 wg_view_shape b_shape) {
    const wg_buf *bufs[2] = { a, b };
    if (int rc = check_common("OpAssign", ctx, dtype, bufs, 2)) return rc;
    if ((int)op < 0 || (int)op > 4) return wg_set_error(WG_ERR_INVALID_ARG, "OpAssign: unknown variant %d", (int)op);
    // op_assign.rs:82-86
    if (a_shape.size[0] != b_shape.size[0])
        return wg_set_error(WG_ERR_DIM_MISMATCH, "Op-assign: dimension mismatch. (a has %u rows, b has %u)", a_shape.size[0],
                            b_shape.size[0]);
    if (a->bytes == 0 || b->bytes == 0) return WG_OK; // kernel.rs:111-123
    const uint32_t n = a_shape.size[0];
    if (n == 0) return WG_OK; // kernel.rs:144
    // op_assign.wgsl:43-45: a[offset_a + i], b[offset_b + i]
    const View va = { n, 1, 1, 1, 1, a_shape.offset }, vb = { n, 1, 1, 1, 1, b_shape.offset };
    if (int rc = check_bounds("OpAssign", "a", va, a, dtype)) return rc;
    if (int rc = check_bounds("OpAssign", "b", vb, b, dtype)) return rc;
    WG_HIP_TRY(hipSetDevice(ctx->device));
    return wgk_op_assign(ctx, (int)op, dtype, (void *)elem_ptr(a, va.offset, dtype), elem_ptr(b, vb.offset, dtype), n);
}

int wg_axpy(wg_ctx *ctx, float alpha, wg_dtype dtype, wg_buf *y, wg_view_shape y_shape, const wg_buf *x, wg_view_shape x_shape) {
    const wg_buf *bufs[2] = { y, x };
    if (int rc = check_common("Axpy", ctx, dtype, bufs, 2)) return rc;
    if (y_shape.size[0] != x_shape.size[0])
        return wg_set_error(WG_ERR_DIM_MISMATCH, "Axpy: dimension mismatch. (y has %u rows, x has %u)", y_shape.size[0], x_shape.size[0]);
    if (y->bytes == 0 || x->bytes == 0) return WG_OK;
    const uint32_t n = y_shape.size[0];
    if (n == 0) return WG_OK;
    const View vy = { n, 1, 1, 1, 1, y_shape.offset }, vx = { n, 1, 1, 1, 1, x_shape.offset };
    if (int rc = check_bounds("Axpy", "y", vy, y, dtype)) return rc;
    if (int rc = check_bounds("Axpy", "x", vx, x, dtype)) return rc;
    WG_HIP_TRY(hipSetDevice(ctx->device));
    return wgk_op_assign(ctx, 5 /* axpy */, dtype, (void *)elem_ptr(y, vy.offset, dtype), elem_ptr(x, vx.offset, dtype), n, alpha);
}

int wg_copy_view(wg_ctx *ctx, wg_dtype dtype, wg_buf *dst, wg_view_shape dst_shape, const wg_buf *src, wg_view_shape src_shape) {
    const wg_buf *bufs[2] = { dst, src };
    if (int rc = check_common("CopyView", ctx, dtype, bufs, 2)) return rc;
    const View d = mk(dst_shape), s = mk(src_shape);
    if (d.mats != s.mats) return wg_set_error(WG_ERR_DIM_MISMATCH, "CopyView: dimension mismatch. (dst has %u matrices, src has %u)", d.mats, s.mats);
    if (dst->bytes == 0 || d.rows == 0 || d.cols == 0 || d.mats == 0) return WG_OK;
    if (int rc = check_bounds("CopyView", "dst", d, dst, dtype)) return rc;
    const bool empty_src = src->bytes == 0 || s.rows == 0 || s.cols == 0;
    if (!empty_src)
        if (int rc = check_bounds("CopyView", "src", s, src, dtype)) return rc;
    WG_HIP_TRY(hipSetDevice(ctx->device));
    return wgk_stage_copy(ctx, dtype, (void *)elem_ptr(dst, d.offset, dtype), d.stride, d.stride_mat, d.rows, d.cols, empty_src ? dst->ptr : elem_ptr(src, s.offset, dtype),
                          s.stride, s.stride_mat, empty_src ? 0u : s.rows, empty_src ? 0u : s.cols, d.mats);
}

// ---------------------------------------------------------------------------------------------------------------
// ROW_MAJOR operator surface. The reference's `Shape` is row-major when its shaders are composed with
// `row_major_shader_defs()` (shape.rs:11-15; shape.wgsl:49-57: index = t*stride_mat + offset + i*stride + j). A row-major
// R x C view is the same memory as the column-major C x R view with the same stride, so:
//   out = m1 m2      <=>  out^T = m2^T m1^T      : the column-major Gemm on the re-labelled views, operands swapped;
//   out = m1^T m2    <=>  out^T = m2^T (m1^T)^T  : needs the second operand transposed in memory: one HBM-bound transpose
//                                                  (transpose.hip) into a scratch buffer, then the column-major Gemm;
//   out = m v        <=>  column-major GemvTr on the re-labelled matrix;   out = m^T v  <=>  column-major Gemv.
// Vector views (Reduce, OpAssign) index `offset + i` in both orderings: nothing to do.
// ---------------------------------------------------------------------------------------------------------------
static wg_view_shape relabel(wg_view_shape s) {
    const uint32_t r = s.size[0];
    s.size[0] = s.size[1];
    s.size[1] = r;
    return s;
}

int wg_gemm_rm(wg_ctx *ctx, wg_gemm_variant variant, wg_dtype dtype, wg_buf *out, wg_view_shape out_shape, const wg_buf *m1,
               wg_view_shape m1_shape, const wg_buf *m2, wg_view_shape m2_shape) {
    const wg_buf *bufs[3] = { out, m1, m2 };
    if (int rc = check_common("Gemm", ctx, dtype, bufs, 3)) return rc;
    if ((int)variant < 0 || (int)variant > 3) return wg_set_error(WG_ERR_INVALID_ARG, "Gemm: unknown variant %d", (int)variant);
    const bool tr = variant == WG_GEMM_TR || variant == WG_GEMM_TR_FAST;
    const View o = mk(out_shape), a = mk(m1_shape), b = mk(m2_shape);
    const uint32_t m_rows = tr ? a.cols : a.rows, m_cols = tr ? a.rows : a.cols; // gemm.rs:81-96, on the row-major shapes
    if (m_cols != b.rows || m_rows != o.rows || o.cols != b.cols || o.mats != a.mats || o.mats != b.mats)
        return wg_set_error(WG_ERR_DIM_MISMATCH,
                            "Gemm: dimension mismatch. (out [%u,%u,%u], m1 [%u,%u,%u]%s, m2 [%u,%u,%u])", o.rows, o.cols, o.mats,
                            a.rows, a.cols, a.mats, tr ? "^T" : "", b.rows, b.cols, b.mats);
    if (!tr) return wg_gemm_ex(ctx, WG_GEMM, dtype, 1.f, 0.f, out, relabel(out_shape), m2, relabel(m2_shape), m1, relabel(m1_shape));

    // m1 is K x M row-major == column-major M x K (ld = stride); the column-major Gemm needs it as K x M column-major
    if (out->bytes == 0 || m1->bytes == 0 || m2->bytes == 0 || o.rows == 0 || o.cols == 0 || o.mats == 0) return WG_OK;
    const View a_cm = mk(relabel(m1_shape)); // M x K column-major
    if (int rc = check_bounds("Gemm", "m1", a_cm, m1, dtype)) return rc;
    const size_t es = dtype == WG_F32 ? 4 : 2;
    const uint32_t K = a.rows, M = a.cols;
    WG_HIP_TRY(hipSetDevice(ctx->device));
    // From about a round of tiles on: the kernels that take m1 where it lies (gemm_f16_nt.hip; gemm_f32.hip's B_NC tile bodies) -- out^T (N x M) = m2^T (N x K,
    // contiguous along N) * m1 (K x M, contiguous along M); no scratch, no extra pass over m1. WG_TUNE_RM_TR_NATIVE = 0 forces the copy below (tests compare the
    // two ways), 1 the kernels at any size.
    if (ctx->tuning[WG_TUNE_RM_TR_NATIVE] != 0) {
        const View o_cm = mk(relabel(out_shape)), b_cm = mk(relabel(m2_shape)); // N x M and N x K column-major
        const uint32_t tile_n = dtype == WG_F16 ? 256u : 128u;                   // 256 x 256 (f16) / 256 x 128 (f32) tiles
        const uint64_t tiles = (uint64_t)((o_cm.rows + 255u) / 256u) * ((o_cm.cols + tile_n - 1u) / tile_n) * o_cm.mats;
        const uint64_t cus = (uint64_t)(ctx->compute_units > 0 ? ctx->compute_units : 256);
        // (f32: the copy path's launcher cuts K / takes small tiles below a round of big tiles, which this launch does not: only from a full round on.
        //  f16: the native launcher has the mid-size tiles too (gemm_f16_t128.hip's B_NC instances): from half a round of 128 x 128 tiles on -- below that the copy
        //  path's launcher would cut K over the idle CUs, which the native kernels do not)
        const uint64_t tiles128 = (uint64_t)((o_cm.rows + 127u) / 128u) * ((o_cm.cols + 127u) / 128u) * o_cm.mats;
        if (ctx->tuning[WG_TUNE_RM_TR_NATIVE] == 1 || (dtype == WG_F16 ? 2u * tiles128 >= cus : tiles >= cus)) {
            if (int rc = check_bounds("Gemm", "m2", b_cm, m2, dtype)) return rc;
            if (int rc = check_bounds("Gemm", "out", o_cm, out, dtype)) return rc;
            const wgk_mat A = { elem_ptr(m2, b_cm.offset, dtype), b_cm.stride, b_cm.stride_mat }, B = { elem_ptr(m1, a_cm.offset, dtype), a_cm.stride, a_cm.stride_mat };
            void *o = (void *)elem_ptr(out, o_cm.offset, dtype);
            const int rc = dtype == WG_F16 ? wgk_gemm_f16_nt(ctx, o_cm.rows, o_cm.cols, K, o_cm.mats, (__half *)o, o_cm.stride, o_cm.stride_mat, A, B)
                                           : wgk_gemm_f32_nt(ctx, o_cm.rows, o_cm.cols, K, o_cm.mats, (float *)o, o_cm.stride, o_cm.stride_mat, A, B);
            if (rc != WG_ERR_UNSUPPORTED) return rc;
        }
    }
    void *ws = nullptr;
    if (int rc = wg_ctx_tr_workspace(ctx, (size_t)K * M * a.mats * es, &ws)) return rc;
    if (int rc = wgk_transpose(ctx, dtype, M, K, a.mats, elem_ptr(m1, a_cm.offset, dtype), a_cm.stride, a_cm.stride_mat, ws, K, (uint64_t)K * M))
        return rc;
    wg_buf tmp;
    tmp.ctx = ctx; tmp.ptr = ws; tmp.bytes = (size_t)K * M * a.mats * es; tmp.usage = 0; tmp.owned = false; tmp.host_pinned = false;
    wg_view_shape ts;
    ts.size[0] = K; ts.size[1] = M; ts.size[2] = a.mats; ts.stride = K; ts.stride_mat = K * M; ts.offset = 0;
    return wg_gemm_ex(ctx, WG_GEMM, dtype, 1.f, 0.f, out, relabel(out_shape), m2, relabel(m2_shape), &tmp, ts);
}

int wg_gemv_rm(wg_ctx *ctx, wg_gemv_variant variant, wg_dtype dtype, wg_buf *out, wg_view_shape out_shape, const wg_buf *m,
               wg_view_shape m_shape, const wg_buf *v, wg_view_shape v_shape) {
    const wg_buf *bufs[3] = { out, m, v };
    if (int rc = check_common("Gemv", ctx, dtype, bufs, 3)) return rc;
    if ((int)variant < 0 || (int)variant > 3) return wg_set_error(WG_ERR_INVALID_ARG, "Gemv: unknown variant %d", (int)variant);
    const bool tr = variant == WG_GEMV_TR || variant == WG_GEMV_TR_FAST;
    const View o = mk(out_shape), mm = mk(m_shape), vv = mk(v_shape);
    const uint32_t m_rows = tr ? mm.cols : mm.rows, m_cols = tr ? mm.rows : mm.cols; // gemv.rs:79-91, on the row-major shapes
    if (m_cols != vv.rows || m_rows != o.rows)
        return wg_set_error(WG_ERR_DIM_MISMATCH, "Gemv: dimension mismatch. (out [%u,%u,%u], m [%u,%u,%u]%s, v [%u,%u,%u])", o.rows,
                            o.cols, o.mats, mm.rows, mm.cols, mm.mats, tr ? "^T" : "", vv.rows, vv.cols, vv.mats);
    if ((variant == WG_GEMV_FAST || variant == WG_GEMV_TR_FAST) && o.rows % 4u != 0)
        return wg_set_error(WG_ERR_PRECONDITION, "Gemv: assertion `left == right` failed (out_nrows %% 4 == 0, gemv.rs:122): out has %u rows",
                            o.rows);
    // Several right-hand sides (grid.y of gemv.wgsl:40,46,62 with the row-major `im`, shape.wgsl:49-57): a row-major matrix of
    // right-hand sides has its columns 1 element apart, which no GEMV kernel streams -- but out (R x n) = op(m) * v (C x n) on row-major
    // views IS the row-major Gemm / GemmTr, i.e. the column-major product out^T (n x R) = v^T * op(m)^T with few rows (the few-row and
    // few-column MFMA kernels). The matrix is shared by the columns and the batch count comes from `out`, as in wg_gemv.
    if (o.cols > 1 || vv.cols > 1) {
        wg_view_shape os = out_shape, ms = m_shape, vs = v_shape;
        ms.size[2] = o.mats;
        vs.size[1] = o.cols;
        vs.size[2] = o.mats;
        return wg_gemm_rm(ctx, tr ? WG_GEMM_TR : WG_GEMM, dtype, out, os, m, ms, v, vs);
    }
    return wg_gemv(ctx, tr ? WG_GEMV : WG_GEMV_TR, dtype, out, out_shape, m, relabel(m_shape), v, v_shape);
}

} // extern "C"
