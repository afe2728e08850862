/*
 * wgsl_oracle.c -- CPU restatement of the wgebra dense linear-algebra WGSL kernels.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (wgmath_amd/, libwgebra_hip.so) never links, imports or calls it.
 *
 * What it restates (all paths relative to /root/reference/crates/wgebra/src/linalg/):
 *   shape.wgsl:36-47,60-66   index math (iv / it / im / with_vec4_elts / div_ceil4)
 *   gemm.wgsl:28-78          gemm_fast      gemm.wgsl:80-113   gemm
 *   gemm.wgsl:115-148        gemm_tr        gemm.wgsl:150-200  gemm_tr_fast
 *   gemv.wgsl:28-65          gemv_fast      gemv.wgsl:67-90    gemv
 *   gemv.wgsl:92-115         gemv_tr        gemv.wgsl:117-155  gemv_tr_fast
 *   reduce.wgsl:12-96        main (+ the five op tables of reduce.rs:30-58)
 *   op_assign.wgsl:14-47     main (+ op table of op_assign.rs:28-38)
 * and the host-side validation / variant->grid mapping / fallbacks of
 *   gemm.rs:75-126, gemv.rs:74-136, reduce.rs:100-113, op_assign.rs:79-94.
 *
 * Each WGSL workgroup/invocation is emulated literally: the same loops, the same u32
 * index arithmetic (wrap-around included), the same per-lane accumulation followed by
 * the same shared-memory tree.  One OpenMP work item per workgroup (fast variants) or
 * per invocation (naive variants), which is also how the cpu_baseline leg times it.
 *
 * PARITY STATUS: "parity unpinned" against a real wgpu/naga EXECUTION for Gemm/Gemv/Reduce:
 * the reference (Rust + wgpu + naga) cannot be built or run in this environment and its
 * own tests hold no golden vectors (unseeded random inputs vs nalgebra, abs eps 1e-3).
 * It is pinned to (a) the reference's OWN SHADER TEXT: tests/golden/wgsl_exec_*.npz hold the
 * outputs of crates/wgebra/src/linalg/*.wgsl executed as they are by oracle/wgsl_exec.py (a
 * small WGSL-subset executor; generator tests/golden/make_wgsl_golden.py) for all ten entry
 * points, and this file reproduces them BIT FOR BIT (tests/test_oracle.py::
 * test_restatement_matches_executed_wgsl_*); (b) the reference's own test procedure and bars,
 * reproduced literally in tests/test_oracle.py (gemm.rs:149-200, gemv.rs:158-195,
 * reduce.rs:143-177); (c) the one deterministic known-answer test the reference holds,
 * gpu_op_assign (op_assign.rs:110-155); (d) an independently written NumPy restatement
 * (oracle/wgsl_oracle.py) that must agree bit-for-bit. What stays unpinned is only what WGSL
 * leaves to the implementation (next paragraph).
 *
 * Unspecified-by-WGSL choices made here (and in the NumPy twin), stated once:
 *   - mat4x4*vec4 and mat4x4*mat4x4 are evaluated as  ((c0*v.x + c1*v.y) + c2*v.z) + c3*v.w
 *     with separate IEEE multiply and add (compile with -ffp-contract=off); naga/the driver
 *     may associate differently or contract to FMA, so GEMM/GEMV/SqNorm parity is
 *     tolerance-based, never bitwise against a real GPU run.
 *   - min/max are the IEEE fminf/fmaxf on non-NaN data (NaN behaviour is
 *     implementation-defined in WGSL; fixtures exclude NaN).
 *
 * Out-of-bounds: the reference disables runtime bounds checks (wgcore utils.rs:11-19), so an
 * OOB access is undefined there.  Here every access is checked; an OOB access makes the call
 * return WGO_ERR_OOB instead of invoking UB.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define WGO_OK 0
#define WGO_ERR_DIM (-1)     /* the Rust assert_eq!(..., "dimension mismatch.") would panic */
#define WGO_ERR_OOB (-2)     /* the WGSL kernel would read/write outside a buffer */
#define WGO_ERR_ASSERT (-3)  /* another Rust-side assert would panic (gemv.rs:122) */
#define WGO_ERR_ARG (-4)

/* shape.wgsl:10-33 == wgcore shapes.rs:9-21 (ViewShape, repr(C), 24 bytes). */
typedef struct {
    uint32_t nrows, ncols, nmats, stride, stride_mat, offset;
} wgo_shape;

typedef struct { float e[4]; } vec4;
typedef struct { vec4 c[4]; } mat4; /* column vectors, like WGSL mat4x4<f32> */

/* shape.wgsl:40-42 */
static inline uint32_t div_ceil4(uint32_t a) { return (a + 3u) / 4u; }
/* shape.wgsl:36-38 */
static inline uint32_t sh_iv(wgo_shape s, uint32_t i) { return s.offset + i; }
/* shape.wgsl:60-62 (column-major branch; ROW_MAJOR is never defined for these kernels) */
static inline uint32_t sh_im(wgo_shape s, uint32_t i, uint32_t j) { return s.offset + i + j * s.stride; }
/* shape.wgsl:45-47 */
static inline uint32_t sh_it(wgo_shape s, uint32_t i, uint32_t j, uint32_t t) { return t * s.stride_mat + sh_im(s, i, j); }
/* shape.wgsl:64-66 */
static inline wgo_shape with_vec4_elts(wgo_shape s) {
    wgo_shape r = { div_ceil4(s.nrows), s.ncols, s.nmats, div_ceil4(s.stride), div_ceil4(s.stride_mat), s.offset / 4u };
    return r;
}

/* A storage buffer bound as array<vec4<f32>>: n4 = floor(len_f32 / 4) addressable vec4s. */
typedef struct { float *p; uint64_t n4; } vbuf;

static inline vec4 vload(vbuf b, uint32_t idx, int *oob) {
    vec4 r = {{0.f, 0.f, 0.f, 0.f}};
    if ((uint64_t)idx >= b.n4) { *oob = 1; return r; }
    memcpy(&r, b.p + 4ull * idx, sizeof r);
    return r;
}
static inline void vstore(vbuf b, uint32_t idx, vec4 v, int *oob) {
    if ((uint64_t)idx >= b.n4) { *oob = 1; return; }
    memcpy(b.p + 4ull * idx, &v, sizeof v);
}

static inline vec4 v_zero(void) { vec4 r = {{0.f, 0.f, 0.f, 0.f}}; return r; }
static inline vec4 v_add(vec4 a, vec4 b) { vec4 r; for (int i = 0; i < 4; ++i) r.e[i] = a.e[i] + b.e[i]; return r; }
static inline mat4 m_zero(void) { mat4 r; for (int j = 0; j < 4; ++j) r.c[j] = v_zero(); return r; }
static inline mat4 m_add(mat4 a, mat4 b) { mat4 r; for (int j = 0; j < 4; ++j) r.c[j] = v_add(a.c[j], b.c[j]); return r; }
/* mat4x4 * vec4 = sum_k column_k * v[k], left to right, no contraction. */
static inline vec4 m_mulv(mat4 a, vec4 v) {
    vec4 r;
    for (int i = 0; i < 4; ++i) {
        float t = a.c[0].e[i] * v.e[0];
        t = t + a.c[1].e[i] * v.e[1];
        t = t + a.c[2].e[i] * v.e[2];
        t = t + a.c[3].e[i] * v.e[3];
        r.e[i] = t;
    }
    return r;
}
/* mat4x4 * mat4x4: result column j = a * b.column_j */
static inline mat4 m_mulm(mat4 a, mat4 b) { mat4 r; for (int j = 0; j < 4; ++j) r.c[j] = m_mulv(a, b.c[j]); return r; }
static inline mat4 m_transpose(mat4 a) {
    mat4 r;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) r.c[j].e[i] = a.c[i].e[j];
    return r;
}
/* the "ia, ib = ia + stride, ic, id" 4-column gather every kernel does */
static inline mat4 load_submat(vbuf b, uint32_t ia, uint32_t stride, int *oob) {
    mat4 r;
    r.c[0] = vload(b, ia, oob);
    r.c[1] = vload(b, ia + stride, oob);
    r.c[2] = vload(b, ia + stride + stride, oob);
    r.c[3] = vload(b, ia + stride + stride + stride, oob);
    return r;
}
static inline void store_submat(vbuf b, uint32_t i_out, uint32_t stride, mat4 m, int *oob) {
    vstore(b, i_out, m.c[0], oob);
    vstore(b, i_out + stride, m.c[1], oob);
    vstore(b, i_out + stride * 2u, m.c[2], oob);
    vstore(b, i_out + stride * 3u, m.c[3], oob);
}

/* ------------------------------------------------------------------------------------- */
/*                                         GEMM                                          */
/* ------------------------------------------------------------------------------------- */
enum { WGO_GEMM = 0, WGO_GEMM_FAST = 1, WGO_GEMM_TR = 2, WGO_GEMM_TR_FAST = 3 }; /* gemm.rs:26-35 */

#define GEMM_WG 64u /* gemm.wgsl:16 */

/* gemm.wgsl:80-113, one invocation (global id x, matrix y). */
static void k_gemm(uint32_t gx, uint32_t gy, wgo_shape so, wgo_shape s1, wgo_shape s2,
                   vbuf out, vbuf m1, vbuf m2, int *oob) {
    s1 = with_vec4_elts(s1); s2 = with_vec4_elts(s2); so = with_vec4_elts(so);
    if (gx < s1.nrows) {
        for (uint32_t k = 0; k < s2.ncols; k += 4u) {
            mat4 sum = m_zero();
            for (uint32_t j = 0; j < s1.ncols; j += 4u) {
                mat4 a = load_submat(m1, sh_it(s1, gx, j, gy), s1.stride, oob);
                mat4 b = load_submat(m2, sh_it(s2, j / 4u, k, gy), s2.stride, oob);
                sum = m_add(sum, m_mulm(a, b));
            }
            store_submat(out, sh_it(so, gx, k, gy), so.stride, sum, oob);
        }
    }
}

/* gemm.wgsl:115-148, one invocation. */
static void k_gemm_tr(uint32_t gx, uint32_t gy, wgo_shape so, wgo_shape s1, wgo_shape s2,
                      vbuf out, vbuf m1, vbuf m2, int *oob) {
    s1 = with_vec4_elts(s1); s2 = with_vec4_elts(s2); so = with_vec4_elts(so);
    if (gx < (s1.ncols + 3u) / 4u) {
        for (uint32_t k = 0; k < s2.ncols; k += 4u) {
            mat4 sum = m_zero();
            for (uint32_t j = 0; j < s1.nrows; j++) {
                mat4 a = load_submat(m1, sh_it(s1, j, gx * 4u, gy), s1.stride, oob);
                mat4 b = load_submat(m2, sh_it(s2, j, k, gy), s2.stride, oob);
                sum = m_add(sum, m_mulm(m_transpose(a), b));
            }
            store_submat(out, sh_it(so, gx, k, gy), so.stride, sum, oob);
        }
    }
}

/* The shared-memory tree of gemm.wgsl:21-26 + :58-63 / :180-185 on 64 mat4 partials. */
static inline void tree64(mat4 *sketch) {
    for (uint32_t stride = 32u; stride >= 1u; stride >>= 1)
        for (uint32_t idx = 0; idx < stride; ++idx) /* "if index < stride" lanes; barrier after */
            sketch[idx] = m_add(sketch[idx], sketch[idx + stride]);
}

/* gemm.wgsl:28-78, one workgroup (id x = 4-row block, y = matrix), all 64 lanes. */
static void k_gemm_fast(uint32_t wx, uint32_t wy, wgo_shape so, wgo_shape s1, wgo_shape s2,
                        vbuf out, vbuf m1, vbuf m2, int *oob) {
    s1 = with_vec4_elts(s1); s2 = with_vec4_elts(s2); so = with_vec4_elts(so);
    mat4 sketch[GEMM_WG];
    for (uint32_t k = 0; k < s2.ncols; k += 4u) {
        for (uint32_t lane = 0; lane < GEMM_WG; ++lane) {
            mat4 sum = m_zero();
            for (uint32_t j = 0; j < s1.ncols; j += 4u * GEMM_WG) {
                mat4 a = load_submat(m1, sh_it(s1, wx, j + lane * 4u, wy), s1.stride, oob);
                mat4 b = load_submat(m2, sh_it(s2, j / 4u + lane, k, wy), s2.stride, oob);
                sum = m_add(sum, m_mulm(a, b));
            }
            sketch[lane] = sum;
        }
        tree64(sketch);
        store_submat(out, sh_it(so, wx, k, wy), so.stride, sketch[0], oob); /* lane 0 only */
    }
}

/* gemm.wgsl:150-200, one workgroup. */
static void k_gemm_tr_fast(uint32_t wx, uint32_t wy, wgo_shape so, wgo_shape s1, wgo_shape s2,
                           vbuf out, vbuf m1, vbuf m2, int *oob) {
    s1 = with_vec4_elts(s1); s2 = with_vec4_elts(s2); so = with_vec4_elts(so);
    mat4 sketch[GEMM_WG];
    for (uint32_t k = 0; k < s2.ncols; k += 4u) {
        for (uint32_t lane = 0; lane < GEMM_WG; ++lane) {
            mat4 sum = m_zero();
            for (uint32_t j = 0; j < s1.nrows; j += GEMM_WG) {
                mat4 a = load_submat(m1, sh_it(s1, j + lane, wx * 4u, wy), s1.stride, oob);
                mat4 b = load_submat(m2, sh_it(s2, j + lane, k, wy), s2.stride, oob);
                sum = m_add(sum, m_mulm(m_transpose(a), b));
            }
            sketch[lane] = sum;
        }
        tree64(sketch);
        store_submat(out, sh_it(so, wx, k, wy), so.stride, sketch[0], oob);
    }
}

static inline uint32_t udiv_ceil(uint32_t a, uint32_t b) { return a / b + (a % b != 0u); }

/*
 * Gemm::dispatch_generic (gemm.rs:65-127).  Buffers are f32 arrays of the given lengths
 * (in f32 elements).  `wg_begin/wg_end` restrict the emulated grid.x range (workgroups) so the
 * cpu_baseline leg can time a bounded sample; pass 0, UINT32_MAX for the whole grid.
 */
int wgo_gemm(int variant, float *out, uint64_t out_len, wgo_shape so,
             const float *m1, uint64_t m1_len, wgo_shape s1,
             const float *m2, uint64_t m2_len, wgo_shape s2,
             uint32_t wg_begin, uint32_t wg_end) {
    if (variant < 0 || variant > 3) return WGO_ERR_ARG;
    const int tr = (variant == WGO_GEMM_TR || variant == WGO_GEMM_TR_FAST);
    const int fast = (variant == WGO_GEMM_FAST || variant == WGO_GEMM_TR_FAST);
    const uint32_t out_rows = so.nrows, out_cols = so.ncols, out_mats = so.nmats;
    /* gemm.rs:81-96 */
    const uint32_t m_rows = tr ? s1.ncols : s1.nrows;
    const uint32_t m_cols = tr ? s1.nrows : s1.ncols;
    if (m_cols != s2.nrows || m_rows != out_rows || out_cols != s2.ncols ||
        out_mats != s1.nmats || out_mats != s2.nmats)
        return WGO_ERR_DIM;
    /* kernel.rs:111-123: a zero-sized bound buffer => the dispatch is silently skipped */
    if (out_len == 0 || m1_len == 0 || m2_len == 0) return WGO_OK;
    /* gemm.rs:109-115 */
    uint32_t grid_x = fast ? udiv_ceil(out_rows, 4u) : udiv_ceil(out_rows, 64u);
    /* kernel.rs:144: any zero grid dimension => skipped */
    if (grid_x == 0 || out_mats == 0) return WGO_OK;

    vbuf vo = { out, out_len / 4u }, v1 = { (float *)m1, m1_len / 4u }, v2 = { (float *)m2, m2_len / 4u };
    if (wg_end > grid_x) wg_end = grid_x;
    int oob_any = 0;
    for (uint32_t wy = 0; wy < out_mats; ++wy) {
        if (fast) {
#pragma omp parallel for schedule(dynamic, 4) reduction(| : oob_any)
            for (int64_t wx = wg_begin; wx < (int64_t)wg_end; ++wx) {
                int oob = 0;
                if (tr) k_gemm_tr_fast((uint32_t)wx, wy, so, s1, s2, vo, v1, v2, &oob);
                else k_gemm_fast((uint32_t)wx, wy, so, s1, s2, vo, v1, v2, &oob);
                oob_any |= oob;
            }
        } else {
            /* 64 invocations per workgroup; invocation id = wx*64 + lane */
#pragma omp parallel for schedule(dynamic, 16) reduction(| : oob_any)
            for (int64_t gx = (int64_t)wg_begin * 64; gx < (int64_t)wg_end * 64; ++gx) {
                int oob = 0;
                if (tr) k_gemm_tr((uint32_t)gx, wy, so, s1, s2, vo, v1, v2, &oob);
                else k_gemm((uint32_t)gx, wy, so, s1, s2, vo, v1, v2, &oob);
                oob_any |= oob;
            }
        }
    }
    return oob_any ? WGO_ERR_OOB : WGO_OK;
}

/* ------------------------------------------------------------------------------------- */
/*                                         GEMV                                          */
/* ------------------------------------------------------------------------------------- */
enum { WGO_GEMV = 0, WGO_GEMV_FAST = 1, WGO_GEMV_TR = 2, WGO_GEMV_TR_FAST = 3 }; /* gemv.rs:25-34 */
#define GEMV_WG 32u /* gemv.wgsl:17 */

/* gemv.wgsl:67-90 */
static void k_gemv(uint32_t gx, uint32_t gy, uint32_t gz, wgo_shape so, wgo_shape sm, wgo_shape sv,
                   vbuf out, vbuf m, vbuf v, int *oob) {
    sm = with_vec4_elts(sm); sv = with_vec4_elts(sv); so = with_vec4_elts(so);
    if (gx < sm.nrows) {
        vec4 sum = v_zero();
        for (uint32_t j = 0; j < sm.ncols; j += 4u) {
            mat4 a = load_submat(m, sh_it(sm, gx, j, gz), sm.stride, oob);
            vec4 x = vload(v, sh_it(sv, j / 4u, gy, gz), oob);
            sum = v_add(sum, m_mulv(a, x));
        }
        vstore(out, sh_it(so, gx, gy, gz), sum, oob);
    }
}

/* gemv.wgsl:92-115 */
static void k_gemv_tr(uint32_t gx, uint32_t gy, uint32_t gz, wgo_shape so, wgo_shape sm, wgo_shape sv,
                      vbuf out, vbuf m, vbuf v, int *oob) {
    sm = with_vec4_elts(sm); sv = with_vec4_elts(sv); so = with_vec4_elts(so);
    if (gx < (sm.ncols + 3u) / 4u) {
        vec4 sum = v_zero();
        for (uint32_t j = 0; j < sm.nrows; j++) {
            mat4 a = load_submat(m, sh_it(sm, j, gx * 4u, gz), sm.stride, oob);
            vec4 x = vload(v, sh_it(sv, j, gy, gz), oob);
            sum = v_add(sum, m_mulv(m_transpose(a), x));
        }
        vstore(out, sh_it(so, gx, gy, gz), sum, oob);
    }
}

/* gemv.wgsl:21-26 + :55-59: strides 16,8,4,2,1 on 32 vec4 partials (the 32-stride step is commented out) */
static inline void tree32(vec4 *sketch) {
    for (uint32_t stride = 16u; stride >= 1u; stride >>= 1)
        for (uint32_t idx = 0; idx < stride; ++idx)
            sketch[idx] = v_add(sketch[idx], sketch[idx + stride]);
}

/* gemv.wgsl:28-65 */
static void k_gemv_fast(uint32_t wx, uint32_t wy, uint32_t wz, wgo_shape so, wgo_shape sm, wgo_shape sv,
                        vbuf out, vbuf m, vbuf v, int *oob) {
    sm = with_vec4_elts(sm); sv = with_vec4_elts(sv); so = with_vec4_elts(so);
    vec4 sketch[GEMV_WG];
    for (uint32_t lane = 0; lane < GEMV_WG; ++lane) {
        vec4 sum = v_zero();
        for (uint32_t j = 0; j < sm.ncols; j += 4u * GEMV_WG) {
            mat4 a = load_submat(m, sh_it(sm, wx, j + lane * 4u, wz), sm.stride, oob);
            vec4 x = vload(v, sh_it(sv, j / 4u + lane, wy, wz), oob);
            sum = v_add(sum, m_mulv(a, x));
        }
        sketch[lane] = sum;
    }
    tree32(sketch);
    vstore(out, sh_it(so, wx, wy, wz), sketch[0], oob);
}

/* gemv.wgsl:117-155 */
static void k_gemv_tr_fast(uint32_t wx, uint32_t wy, uint32_t wz, wgo_shape so, wgo_shape sm, wgo_shape sv,
                           vbuf out, vbuf m, vbuf v, int *oob) {
    sm = with_vec4_elts(sm); sv = with_vec4_elts(sv); so = with_vec4_elts(so);
    vec4 sketch[GEMV_WG];
    for (uint32_t lane = 0; lane < GEMV_WG; ++lane) {
        vec4 sum = v_zero();
        for (uint32_t j = 0; j < sm.nrows; j += GEMV_WG) {
            mat4 a = load_submat(m, sh_it(sm, j + lane, wx * 4u, wz), sm.stride, oob);
            vec4 x = vload(v, sh_it(sv, j + lane, wy, wz), oob);
            sum = v_add(sum, m_mulv(m_transpose(a), x));
        }
        sketch[lane] = sum;
    }
    tree32(sketch);
    vstore(out, sh_it(so, wx, wy, wz), sketch[0], oob);
}

/* Gemv::dispatch_generic (gemv.rs:64-137). */
int wgo_gemv(int variant, float *out, uint64_t out_len, wgo_shape so,
             const float *m, uint64_t m_len, wgo_shape sm,
             const float *v, uint64_t v_len, wgo_shape sv,
             uint32_t wg_begin, uint32_t wg_end) {
    if (variant < 0 || variant > 3) return WGO_ERR_ARG;
    const int tr0 = (variant == WGO_GEMV_TR || variant == WGO_GEMV_TR_FAST);
    const uint32_t out_nrows = so.nrows, out_ncols = so.ncols, out_nmats = so.nmats;
    /* gemv.rs:79-91 -- only these two are checked */
    const uint32_t m_rows = tr0 ? sm.ncols : sm.nrows;
    const uint32_t m_cols = tr0 ? sm.nrows : sm.ncols;
    if (m_cols != sv.nrows || m_rows != out_nrows) return WGO_ERR_DIM;
    /* gemv.rs:99-104: silent fallback */
    if (variant == WGO_GEMV_TR_FAST && sm.nrows % (GEMV_WG * 4u) != 0u) variant = WGO_GEMV_TR;
    const int tr = (variant == WGO_GEMV_TR || variant == WGO_GEMV_TR_FAST);
    const int fast = (variant == WGO_GEMV_FAST || variant == WGO_GEMV_TR_FAST);
    uint32_t grid_x;
    if (fast) {
        if (out_nrows % 4u != 0u) return WGO_ERR_ASSERT; /* gemv.rs:122 assert_eq!(out_nrows % 4, 0) */
        grid_x = udiv_ceil(out_nrows, 4u);
    } else {
        grid_x = udiv_ceil(out_nrows, GEMV_WG);
    }
    if (out_len == 0 || m_len == 0 || v_len == 0) return WGO_OK; /* kernel.rs:111-123 */
    if (grid_x == 0 || out_ncols == 0 || out_nmats == 0) return WGO_OK; /* kernel.rs:144 */

    vbuf vo = { out, out_len / 4u }, vm = { (float *)m, m_len / 4u }, vv = { (float *)v, v_len / 4u };
    if (wg_end > grid_x) wg_end = grid_x;
    int oob_any = 0;
    for (uint32_t wz = 0; wz < out_nmats; ++wz)
        for (uint32_t wy = 0; wy < out_ncols; ++wy) {
            if (fast) {
#pragma omp parallel for schedule(dynamic, 8) reduction(| : oob_any)
                for (int64_t wx = wg_begin; wx < (int64_t)wg_end; ++wx) {
                    int oob = 0;
                    if (tr) k_gemv_tr_fast((uint32_t)wx, wy, wz, so, sm, sv, vo, vm, vv, &oob);
                    else k_gemv_fast((uint32_t)wx, wy, wz, so, sm, sv, vo, vm, vv, &oob);
                    oob_any |= oob;
                }
            } else {
#pragma omp parallel for schedule(dynamic, 32) reduction(| : oob_any)
                for (int64_t gx = (int64_t)wg_begin * GEMV_WG; gx < (int64_t)wg_end * GEMV_WG; ++gx) {
                    int oob = 0;
                    if (tr) k_gemv_tr((uint32_t)gx, wy, wz, so, sm, sv, vo, vm, vv, &oob);
                    else k_gemv((uint32_t)gx, wy, wz, so, sm, sv, vo, vm, vv, &oob);
                    oob_any |= oob;
                }
            }
        }
    return oob_any ? WGO_ERR_OOB : WGO_OK;
}

/* ------------------------------------------------------------------------------------- */
/*                                        REDUCE                                         */
/* ------------------------------------------------------------------------------------- */
enum { WGO_MIN = 0, WGO_MAX = 1, WGO_SUM = 2, WGO_PROD = 3, WGO_SQNORM = 4 }; /* reduce.rs:13-27 */
#define REDUCE_WG 128u /* reduce.wgsl:10 */

/* reduce.wgsl:12-46, selected by reduce.rs:30-58 */
static inline float red_init(int op) {
    switch (op) {
    case WGO_MIN: return 3.4e38f;   /* init_max_f32, reduce.wgsl:40-42: NOT FLT_MAX */
    case WGO_MAX: return -3.4e38f;  /* init_min_f32, reduce.wgsl:44-46 */
    case WGO_PROD: return 1.0f;
    default: return 0.0f;
    }
}
static inline float red_workspace(int op, float acc, float x) {
    switch (op) {
    case WGO_MIN: return fminf(acc, x);
    case WGO_MAX: return fmaxf(acc, x);
    case WGO_SUM: return acc + x;
    case WGO_PROD: return acc * x;
    default: { float sq = x * x; return acc + sq; } /* reduce_sqnorm_f32, no contraction */
    }
}
static inline float red_reduce(int op, float a, float b) {
    switch (op) {
    case WGO_MIN: return fminf(a, b);
    case WGO_MAX: return fmaxf(a, b);
    case WGO_PROD: return a * b;
    default: return a + b; /* Sum, and SqNorm (reduce.rs:55: "reduce_sqnorm only happens in workspace") */
    }
}

/* reduce.wgsl:68-96: the one 128-lane workgroup. `in_len` in f32 elements. */
static int reduce_one(int op, const float *input, uint64_t in_len, wgo_shape shape, float *result) {
    float ws[REDUCE_WG];
    int oob = 0;
    for (uint32_t t = 0; t < REDUCE_WG; ++t) {
        ws[t] = red_init(op);
        for (uint32_t i = t; i < shape.nrows; i += REDUCE_WG) {
            uint32_t idx = sh_iv(shape, i);
            if ((uint64_t)idx >= in_len) { oob = 1; continue; }
            ws[t] = red_workspace(op, ws[t], input[idx]);
            if (i + REDUCE_WG < i) break; /* u32 wrap guard for n near 2^32 */
        }
    }
    for (uint32_t stride = 64u; stride >= 1u; stride >>= 1)
        for (uint32_t t = 0; t < stride; ++t)
            ws[t] = red_reduce(op, ws[t], ws[t + stride]);
    *result = ws[0];
    return oob;
}

/* Reduce::dispatch (reduce.rs:100-113): dispatch(1); result is the GpuScalar's single f32. */
int wgo_reduce(int op, const float *input, uint64_t in_len, wgo_shape shape, float *result, uint64_t result_len) {
    if (op < 0 || op > 4) return WGO_ERR_ARG;
    if (in_len == 0 || result_len == 0) return WGO_OK; /* kernel.rs:111-123 */
    float r;
    int oob = reduce_one(op, input, in_len, shape, &r);
    if (oob) return WGO_ERR_OOB;
    result[0] = r;
    return WGO_OK;
}

/*
 * Batched form used by BASELINE config 4 ("4096 x (65536-vec)"): column c of the col-major view is
 * reduced exactly as Reduce::dispatch would reduce the vector view {size:[nrows,1,1], offset:
 * offset + c*stride (+ t*stride_mat)} -- i.e. ncols*nmats independent reference dispatches.
 * results[c + t*ncols].
 */
int wgo_reduce_batched(int op, const float *input, uint64_t in_len, wgo_shape shape, float *results, uint64_t results_len) {
    if (op < 0 || op > 4) return WGO_ERR_ARG;
    uint64_t nvec = (uint64_t)shape.ncols * shape.nmats;
    if (results_len < nvec) return WGO_ERR_ARG;
    if (in_len == 0 || nvec == 0) return WGO_OK;
    int oob_any = 0;
#pragma omp parallel for schedule(static) reduction(| : oob_any)
    for (int64_t q = 0; q < (int64_t)nvec; ++q) {
        uint32_t c = (uint32_t)(q % shape.ncols), t = (uint32_t)(q / shape.ncols);
        wgo_shape s = { shape.nrows, 1u, 1u, shape.stride, shape.stride_mat,
                        shape.offset + c * shape.stride + t * shape.stride_mat };
        float r;
        oob_any |= reduce_one(op, input, in_len, s, &r);
        results[q] = r;
    }
    return oob_any ? WGO_ERR_OOB : WGO_OK;
}

/* ------------------------------------------------------------------------------------- */
/*                                       OP-ASSIGN                                       */
/* ------------------------------------------------------------------------------------- */
enum { WGO_ADD = 0, WGO_SUB = 1, WGO_MUL = 2, WGO_DIV = 3, WGO_COPY = 4 }; /* op_assign.rs:12-26 */

/* OpAssign::dispatch (op_assign.rs:71-95) + op_assign.wgsl:40-47; grid = ceil(n/64) x 64 lanes. */
int wgo_op_assign(int op, float *a, uint64_t a_len, wgo_shape sa, const float *b, uint64_t b_len, wgo_shape sb) {
    if (op < 0 || op > 4) return WGO_ERR_ARG;
    if (sa.nrows != sb.nrows) return WGO_ERR_DIM; /* op_assign.rs:82-86 */
    if (a_len == 0 || b_len == 0) return WGO_OK;
    const uint32_t n = sa.nrows;
    int oob_any = 0;
#pragma omp parallel for schedule(static) reduction(| : oob_any)
    for (int64_t x = 0; x < (int64_t)n; ++x) { /* "if invocation_id.x < shape_a.nrows" */
        uint32_t ia = sh_iv(sa, (uint32_t)x), ib = sh_iv(sb, (uint32_t)x);
        if ((uint64_t)ia >= a_len || (uint64_t)ib >= b_len) { oob_any |= 1; continue; }
        float av = a[ia], bv = b[ib], r;
        switch (op) {
        case WGO_ADD: r = av + bv; break;
        case WGO_SUB: r = av - bv; break;
        case WGO_MUL: r = av * bv; break;
        case WGO_DIV: r = av / bv; break;
        default: r = bv; break;
        }
        a[ia] = r;
    }
    return oob_any ? WGO_ERR_OOB : WGO_OK;
}

/* Extension checked by the oracle too (no reference counterpart): y[i] = fmaf(alpha, x[i], y[i]), op_assign's indexing. */
int wgo_axpy(float alpha, float *y, uint64_t y_len, wgo_shape sy, const float *x, uint64_t x_len, wgo_shape sx) {
    if (sy.nrows != sx.nrows) return WGO_ERR_DIM;
    if (y_len == 0 || x_len == 0) return WGO_OK;
    int oob_any = 0;
#pragma omp parallel for schedule(static) reduction(| : oob_any)
    for (int64_t i = 0; i < (int64_t)sy.nrows; ++i) {
        uint32_t iy = sh_iv(sy, (uint32_t)i), ix = sh_iv(sx, (uint32_t)i);
        if ((uint64_t)iy >= y_len || (uint64_t)ix >= x_len) { oob_any |= 1; continue; }
        y[iy] = fmaf(alpha, x[ix], y[iy]);
    }
    return oob_any ? WGO_ERR_OOB : WGO_OK;
}

/* ------------------------------------------------------------------------------------- */
int wgo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void wgo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
const char *wgo_version(void) { return "wgsl_oracle 1 (restates wgebra linalg WGSL @ reference 2025-12-05)"; }
