// wgebra_geometry.hpp -- small-matrix geometry for device (HIP) and host code: the functions of the reference's
// `wgebra::geometry` module (crates/wgebra/src/geometry/*.wgsl) as a header of `__host__ __device__`
// inline functions. They are building blocks called from inside other kernels (one matrix / transform per thread), not
// data-parallel operators of their own: there is nothing to tile or stage.
//
// Conventions follow the reference:
//   * matrices are column-major, `m.c[col][row]` == WGSL `m[col][row]`; N = 2, 3, 4;
//   * inv2/inv3/inv4         (inv.wgsl)       : the inverse; an invalid (inf/nan) result for a singular matrix;
//   * cholesky               (cholesky.wgsl)  : L in the lower triangle (diagonal included), the upper triangle keeps the input;
//   * lu                     (lu.wgsl)        : partial pivoting; L (unit diagonal implicit) below, U on and above the diagonal,
//                                               + the row swaps as (ia[i], ib[i]), i < len;
//   * qr                     (qr2/3/4.wgsl)   : nalgebra's Householder QR as ported by the reference; diag(R) >= 0;
//   * symmetric_eigen        (eig2/3/4.wgsl)  : M = V diag(lambda) V^T; N = 2 closed form with eigenvalues ((a+b+sigma)/2, (a+b-sigma)/2) in that
//                                               order; N = 3, 4 tridiagonalisation + Wilkinson-shift implicit QR, eigenvalues unsorted as the
//                                               deflation leaves them;
//   * svd (2, 3)             (svd2/3.wgsl)    : M = U diag(S) Vt; N = 2 closed form through stable_atan2 (S sorted, >= 0); N = 3 the
//                                               Givens-quaternion Jacobi + QR of McAdams et al. (S sorted by magnitude, the last may be < 0);
//   * trig, min_max          (utils/*.wgsl)   : stable_tanh, stable_atan2 (0 on the axis x <= 0, y == 0 and for x == 0), max / amax of vectors and matrices;
//   * Quat, Rot2, Sim2, Sim3 (quat/rot2/sim2/sim3.wgsl): unit quaternion (x, y, z, w), 2-D rotation (cos, sin), similarities
//                                               x -> scale * R x + t, with the reference's function names.
// EVERY function follows the reference's WGSL statement by statement and expression by expression, every product and sum rounded separately (no FMA
// contraction: WGG_EXACT below; `fma(..)` where the WGSL says fma), so that on the same inputs it gives the bits the WGSL text gives when read that
// way (left to right, `dot` / `cross` / `length` / matrix products as their defining formulas, sqrt and division correctly rounded, sin / cos / atan /
// exp correctly rounded): tests/golden/wgsl_exec_geometry.npz and wgsl_exec_decomp.npz, made by running the reference's .wgsl files through the tests'
// WGSL executor, are matched at 0 ulp by the host build and by the HIP kernels (tests/test_geometry.py).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define WGG_FN __host__ __device__ inline
#else
#define WGG_FN inline
#endif
// First statement of a function whose results are pinned to the WGSL text: a * b + c stays a rounded product and a rounded sum whatever
// -ffp-contract the including translation unit was built with. (g++ has no per-function switch: its host builds of the pinned functions are
// compiled with -ffp-contract=off, tests/test_geometry.py.)
#if defined(__clang__)
#define WGG_EXACT _Pragma("clang fp contract(off)")
#else
#define WGG_EXACT
#endif

namespace wgebra {
namespace geometry {

template <int N>
struct Mat {
    float c[N][N]; // c[col][row]
};
template <int N>
struct Vec {
    float v[N];
};
using Mat2 = Mat<2>;
using Mat3 = Mat<3>;
using Mat4 = Mat<4>;

template <int N>
WGG_FN Mat<N> identity() {
    Mat<N> r;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) r.c[j][i] = i == j ? 1.f : 0.f;
    return r;
}
template <int N>
WGG_FN Mat<N> mul(const Mat<N> &a, const Mat<N> &b) {
    Mat<N> r;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) {
            float s = 0.f;
            for (int k = 0; k < N; ++k) s += a.c[k][i] * b.c[j][k];
            r.c[j][i] = s;
        }
    return r;
}
template <int N>
WGG_FN Mat<N> transpose(const Mat<N> &a) {
    Mat<N> r;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) r.c[j][i] = a.c[i][j];
    return r;
}
template <int N>
WGG_FN Vec<N> mul(const Mat<N> &a, const Vec<N> &x) {
    Vec<N> r;
    for (int i = 0; i < N; ++i) {
        float s = 0.f;
        for (int k = 0; k < N; ++k) s += a.c[k][i] * x.v[k];
        r.v[i] = s;
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// WGSL builtins as the functions pinned to the reference's text read them (the fixtures under tests/golden/ fix the same reading)
// ---------------------------------------------------------------------------------------------------------------
// sin / cos / atan / exp: the correctly rounded f32 value (the float64 function rounded once) -- the one result every conformant WGSL implementation's
// error bound contains, and the only one a fixture can pin. These are per-item building blocks, not throughput code: the f64 evaluation costs nothing
// that matters and makes the host build, the HIP build and the executed-WGSL fixtures agree bit for bit.
WGG_FN float w_sin(float x) { return (float)sin((double)x); }
WGG_FN float w_cos(float x) { return (float)cos((double)x); }
WGG_FN float w_atan(float x) { return (float)atan((double)x); }
WGG_FN float w_exp(float x) { return (float)exp((double)x); }
// sign(): 1, -1, and +0 for both zeros (WGSL leaves sign(-0) open; the fixtures use +0)
WGG_FN float w_sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
WGG_FN float w_max(float a, float b) { return a < b ? b : a; }
WGG_FN float w_length2(float x, float y) { WGG_EXACT return sqrtf(x * x + y * y); }
// matN x matN as WGSL evaluates it here: element (row i, column j) = a[0][i] b[j][0] + a[1][i] b[j][1] + ..., left to right from the first product
template <int N>
WGG_FN Mat<N> mul_exact(const Mat<N> &a, const Mat<N> &b) {
    WGG_EXACT
    Mat<N> r;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) {
            float s = a.c[0][i] * b.c[j][0];
            for (int k = 1; k < N; ++k) s = s + a.c[k][i] * b.c[j][k];
            r.c[j][i] = s;
        }
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// wgebra::trig (utils/trig.wgsl:15-38)
// ---------------------------------------------------------------------------------------------------------------
namespace trig {
constexpr float PI = 3.14159265358979323846264338327950288f; // trig.wgsl:4
WGG_FN float stable_tanh(float x) { // trig.wgsl:12-18: both branches are evaluated, the sign of x selects
    WGG_EXACT
    const float exp_neg2x = w_exp(-2.f * x);
    const float exp_pos2x = w_exp(2.f * x);
    const float tanh_pos = (1.f - exp_neg2x) / (1.f + exp_neg2x);
    const float tanh_neg = (exp_pos2x - 1.f) / (exp_pos2x + 1.f);
    return x >= 0.f ? tanh_pos : tanh_neg;
}
WGG_FN float stable_atan2(float y, float x) { // trig.wgsl:25-38: 0 for x == 0 and for x < 0, y == 0 (NOT pi / +-pi/2: the reference's choice)
    WGG_EXACT
    const float ang = w_atan(y / x);
    if (x > 0.f) return ang;
    if (x < 0.f && y > 0.f) return ang + PI;
    if (x < 0.f && y < 0.f) return ang - PI;
    return 0.f;
}
} // namespace trig

// ---------------------------------------------------------------------------------------------------------------
// wgebra::min_max (utils/min_max.wgsl:4-51)
// ---------------------------------------------------------------------------------------------------------------
namespace min_max {
template <int N>
WGG_FN float maxv(const Vec<N> &v) { // max2 / max3 / max4: max(v.x, max(v.y, ..))
    float r = v.v[N - 1];
    for (int k = N - 2; k >= 0; --k) r = w_max(v.v[k], r);
    return r;
}
template <int N>
WGG_FN float maxm(const Mat<N> &m) { // max2x2 / max3x3 / max4x4: the component-wise max of the columns, then maxv
    Vec<N> vm;
    for (int i = 0; i < N; ++i) {
        float r = m.c[N - 1][i];
        for (int k = N - 2; k >= 0; --k) r = w_max(m.c[k][i], r);
        vm.v[i] = r;
    }
    return maxv<N>(vm);
}
template <int N>
WGG_FN float amaxm(const Mat<N> &m) { // amax2x2 / amax3x3 / amax4x4
    Mat<N> a;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) a.c[j][i] = fabsf(m.c[j][i]);
    return maxm<N>(a);
}
WGG_FN float max2(const Vec<2> &v) { return maxv<2>(v); }
WGG_FN float max3(const Vec<3> &v) { return maxv<3>(v); }
WGG_FN float max4(const Vec<4> &v) { return maxv<4>(v); }
WGG_FN float max2x2(const Mat2 &m) { return maxm<2>(m); }
WGG_FN float max3x3(const Mat3 &m) { return maxm<3>(m); }
WGG_FN float max4x4(const Mat4 &m) { return maxm<4>(m); }
WGG_FN float amax2x2(const Mat2 &m) { return amaxm<2>(m); }
WGG_FN float amax3x3(const Mat3 &m) { return amaxm<3>(m); }
WGG_FN float amax4x4(const Mat4 &m) { return amaxm<4>(m); }
} // namespace min_max

// ---------------------------------------------------------------------------------------------------------------
// inverse (inv.wgsl:8-88): adjugate / determinant
// ---------------------------------------------------------------------------------------------------------------
WGG_FN Mat2 inv2(const Mat2 &m) { // inv.wgsl:8-18
    WGG_EXACT
    Mat2 adj;
    adj.c[0][0] = m.c[1][1];
    adj.c[0][1] = -m.c[0][1];
    adj.c[1][0] = -m.c[1][0];
    adj.c[1][1] = m.c[0][0];
    const float det = m.c[0][0] * m.c[1][1] - m.c[1][0] * m.c[0][1];
    const float s = 1.f / det;
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 2; ++i) adj.c[j][i] = adj.c[j][i] * s;
    return adj;
}
WGG_FN Mat3 inv3(const Mat3 &m) { // inv.wgsl:24-43
    WGG_EXACT
    const float(&a)[3][3] = m.c; // a[col][row] == WGSL m[col][row]
    Mat3 adj;
    adj.c[0][0] = (a[1][1] * a[2][2] - a[2][1] * a[1][2]);
    adj.c[1][0] = -(a[1][0] * a[2][2] - a[2][0] * a[1][2]);
    adj.c[2][0] = (a[1][0] * a[2][1] - a[2][0] * a[1][1]);
    adj.c[0][1] = -(a[0][1] * a[2][2] - a[2][1] * a[0][2]);
    adj.c[1][1] = (a[0][0] * a[2][2] - a[2][0] * a[0][2]);
    adj.c[2][1] = -(a[0][0] * a[2][1] - a[2][0] * a[0][1]);
    adj.c[0][2] = (a[0][1] * a[1][2] - a[1][1] * a[0][2]);
    adj.c[1][2] = -(a[0][0] * a[1][2] - a[1][0] * a[0][2]);
    adj.c[2][2] = (a[0][0] * a[1][1] - a[1][0] * a[0][1]);
    const float det = (a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                       a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
    const float s = 1.f / det;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) adj.c[j][i] = adj.c[j][i] * s;
    return adj;
}
WGG_FN Mat4 inv4(const Mat4 &m) { // inv.wgsl:49-88
    WGG_EXACT
    const float(&a)[4][4] = m.c;
    const float sf00 = a[2][2] * a[3][3] - a[3][2] * a[2][3], sf01 = a[2][1] * a[3][3] - a[3][1] * a[2][3];
    const float sf02 = a[2][1] * a[3][2] - a[3][1] * a[2][2], sf03 = a[2][0] * a[3][3] - a[3][0] * a[2][3];
    const float sf04 = a[2][0] * a[3][2] - a[3][0] * a[2][2], sf05 = a[2][0] * a[3][1] - a[3][0] * a[2][1];
    const float sf06 = a[1][2] * a[3][3] - a[3][2] * a[1][3], sf07 = a[1][1] * a[3][3] - a[3][1] * a[1][3];
    const float sf08 = a[1][1] * a[3][2] - a[3][1] * a[1][2], sf09 = a[1][0] * a[3][3] - a[3][0] * a[1][3];
    const float sf10 = a[1][0] * a[3][2] - a[3][0] * a[1][2], sf11 = a[1][1] * a[3][3] - a[3][1] * a[1][3];
    const float sf12 = a[1][0] * a[3][1] - a[3][0] * a[1][1], sf13 = a[1][2] * a[2][3] - a[2][2] * a[1][3];
    const float sf14 = a[1][1] * a[2][3] - a[2][1] * a[1][3], sf15 = a[1][1] * a[2][2] - a[2][1] * a[1][2];
    const float sf16 = a[1][0] * a[2][3] - a[2][0] * a[1][3], sf17 = a[1][0] * a[2][2] - a[2][0] * a[1][2];
    const float sf18 = a[1][0] * a[2][1] - a[2][0] * a[1][1];
    Mat4 adj;
    adj.c[0][0] = (a[1][1] * sf00 - a[1][2] * sf01 + a[1][3] * sf02);
    adj.c[1][0] = -(a[1][0] * sf00 - a[1][2] * sf03 + a[1][3] * sf04);
    adj.c[2][0] = (a[1][0] * sf01 - a[1][1] * sf03 + a[1][3] * sf05);
    adj.c[3][0] = -(a[1][0] * sf02 - a[1][1] * sf04 + a[1][2] * sf05);
    adj.c[0][1] = -(a[0][1] * sf00 - a[0][2] * sf01 + a[0][3] * sf02);
    adj.c[1][1] = (a[0][0] * sf00 - a[0][2] * sf03 + a[0][3] * sf04);
    adj.c[2][1] = -(a[0][0] * sf01 - a[0][1] * sf03 + a[0][3] * sf05);
    adj.c[3][1] = (a[0][0] * sf02 - a[0][1] * sf04 + a[0][2] * sf05);
    adj.c[0][2] = (a[0][1] * sf06 - a[0][2] * sf07 + a[0][3] * sf08);
    adj.c[1][2] = -(a[0][0] * sf06 - a[0][2] * sf09 + a[0][3] * sf10);
    adj.c[2][2] = (a[0][0] * sf11 - a[0][1] * sf09 + a[0][3] * sf12);
    adj.c[3][2] = -(a[0][0] * sf08 - a[0][1] * sf10 + a[0][2] * sf12);
    adj.c[0][3] = -(a[0][1] * sf13 - a[0][2] * sf14 + a[0][3] * sf15);
    adj.c[1][3] = (a[0][0] * sf13 - a[0][2] * sf16 + a[0][3] * sf17);
    adj.c[2][3] = -(a[0][0] * sf14 - a[0][1] * sf16 + a[0][3] * sf18);
    adj.c[3][3] = (a[0][0] * sf15 - a[0][1] * sf17 + a[0][2] * sf18);
    const float det = (a[0][0] * adj.c[0][0] + a[0][1] * adj.c[1][0] + a[0][2] * adj.c[2][0] + a[0][3] * adj.c[3][0]);
    const float s = 1.f / det;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) adj.c[j][i] = adj.c[j][i] * s;
    return adj;
}
template <int N>
WGG_FN Mat<N> inv(const Mat<N> &m);
template <>
WGG_FN Mat2 inv<2>(const Mat2 &m) { return inv2(m); }
template <>
WGG_FN Mat3 inv<3>(const Mat3 &m) { return inv3(m); }
template <>
WGG_FN Mat4 inv<4>(const Mat4 &m) { return inv4(m); }

// ---------------------------------------------------------------------------------------------------------------
// Cholesky (cholesky.wgsl:16-35): lower triangle <- L, upper triangle untouched
// ---------------------------------------------------------------------------------------------------------------
template <int N>
WGG_FN Mat<N> cholesky(const Mat<N> &x) {
    WGG_EXACT
    Mat<N> m = x;
    for (int j = 0; j < N; ++j) {
        for (int k = 0; k < j; ++k) {
            const float factor = -m.c[k][j]; // -L(j, k)
            for (int l = j; l < N; ++l) m.c[j][l] = m.c[j][l] + factor * m.c[k][l];
        }
        const float d = sqrtf(m.c[j][j]);
        m.c[j][j] = d;
        for (int l = j + 1; l < N; ++l) m.c[j][l] /= d;
    }
    return m;
}

// ---------------------------------------------------------------------------------------------------------------
// LU with partial pivoting (lu.wgsl:36-132)
// ---------------------------------------------------------------------------------------------------------------
template <int N>
struct Permutations {
    uint32_t ia[N], ib[N], len;
};
template <int N>
struct LU {
    Mat<N> lu;
    Permutations<N> p;
};
template <int N>
WGG_FN LU<N> lu(const Mat<N> &x) {
    WGG_EXACT
    LU<N> r;
    r.lu = x;
    r.p.len = 0;
    for (int i = 0; i < N; ++i) r.p.ia[i] = r.p.ib[i] = 0;
    Mat<N> &m = r.lu;
    for (int i = 0; i < N; ++i) {
        int piv = i;
        float best = fabsf(m.c[i][i]);
        for (int rr = i + 1; rr < N; ++rr) {
            const float v = fabsf(m.c[i][rr]);
            if (v > best) { best = v; piv = rr; }
        }
        if (best == 0.f) continue; // no non-zero entry in this column
        if (piv != i) {
            r.p.ia[r.p.len] = (uint32_t)i;
            r.p.ib[r.p.len] = (uint32_t)piv;
            ++r.p.len;
            for (int cc = 0; cc < N; ++cc) { const float t = m.c[cc][i]; m.c[cc][i] = m.c[cc][piv]; m.c[cc][piv] = t; }
        }
        const float inv_diag = 1.f / m.c[i][i]; // lu.wgsl:84-99: the pivot's reciprocal, then products (not divisions)
        for (int rr = i + 1; rr < N; ++rr) m.c[i][rr] = m.c[i][rr] * inv_diag;
        for (int cc = i + 1; cc < N; ++cc) {
            const float pivot = m.c[cc][i];
            for (int rr = i + 1; rr < N; ++rr) m.c[cc][rr] = m.c[cc][rr] - pivot * m.c[i][rr];
        }
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// QR (qr2.wgsl:15-107, qr3.wgsl:15-109, qr4.wgsl:15-111): nalgebra's Householder QR as the reference ported it, statement for statement --
// the reflection axes are normalised twice (householder::reflection_axis_mut), each reflection is also applied to its own axis column
// (the `c = i` trip of the loop), Q is assembled backwards from the stored axes, diag(R) = |diag| >= 0.
// ---------------------------------------------------------------------------------------------------------------
template <int N>
struct QR {
    Mat<N> q, r;
};
template <int N>
WGG_FN QR<N> qr(const Mat<N> &x) {
    WGG_EXACT
    Mat<N> m = x;
    float diag[N];
    for (int i = 0; i < N; ++i) diag[i] = 0.f;
    for (int i = 0; i < N; ++i) {
        float axis_sq_norm = 0.f; // the axis is m[i.., i]
        for (int r = i; r < N; ++r) axis_sq_norm = axis_sq_norm + m.c[i][r] * m.c[i][r];
        const float axis_norm = sqrtf(axis_sq_norm);
        const float modulus = fabsf(m.c[i][i]);
        const float signed_norm = w_sign(m.c[i][i]) * axis_norm;
        const float factor = (axis_sq_norm + modulus * axis_norm) * 2.f;
        m.c[i][i] = m.c[i][i] + signed_norm;
        if (factor != 0.f) {
            const float factor_sqrt = sqrtf(factor);
            float norm = 0.f;
            for (int r = i; r < N; ++r) {
                m.c[i][r] = m.c[i][r] / factor_sqrt;
                norm = norm + m.c[i][r] * m.c[i][r];
            }
            norm = sqrtf(norm);
            for (int r = i; r < N; ++r) m.c[i][r] = m.c[i][r] / norm; // renormalisation
            diag[i] = -signed_norm;
            const float sgn = w_sign(diag[i]); // reflect_with_sign on the columns i.. (column i, the axis itself, included)
            for (int c = i; c < N; ++c) {
                const float m_two = -2.f * sgn;
                float f = 0.f;
                for (int r = i; r < N; ++r) f = f + m.c[i][r] * m.c[c][r];
                for (int r = i; r < N; ++r) m.c[c][r] = m_two * f * m.c[i][r] + m.c[c][r] * sgn;
            }
        } else {
            diag[i] = signed_norm;
        }
    }
    QR<N> o;
    o.q = identity<N>(); // QR::q() of nalgebra: the reflections applied to the identity, last one first
    for (int i = N - 1; i >= 0; --i) {
        const float sgn = w_sign(diag[i]);
        for (int c = i; c < N; ++c) {
            const float m_two = -2.f * sgn;
            float f = 0.f;
            for (int r = i; r < N; ++r) f = f + m.c[i][r] * o.q.c[c][r];
            for (int r = i; r < N; ++r) o.q.c[c][r] = m_two * f * m.c[i][r] + o.q.c[c][r] * sgn;
        }
    }
    for (int c = 0; c < N; ++c) // R: the strict upper triangle of m, |diag| on the diagonal, zeros below
        for (int r = 0; r < N; ++r) o.r.c[c][r] = r < c ? m.c[c][r] : (r == c ? fabsf(diag[c]) : 0.f);
    return o;
}

// ---------------------------------------------------------------------------------------------------------------
// Rot2 (rot2.wgsl): 2-D rotation stored as (cos, sin)
// ---------------------------------------------------------------------------------------------------------------
struct Rot2 {
    float cos, sin;
};
namespace rot2 {
WGG_FN Rot2 identity() { return Rot2{ 1.f, 0.f }; }
WGG_FN bool is_valid(Rot2 r) { return r.cos != 0.f || r.sin != 0.f; } // rot2.wgsl:15-17: the zero Rot2 means "no rotation found"
WGG_FN Rot2 fromAngle(float angle) { return Rot2{ w_cos(angle), w_sin(angle) }; }
WGG_FN float angle(Rot2 r) { return trig::stable_atan2(r.sin, r.cos); } // rot2.wgsl:51-53: 0 (not pi) for (cos, sin) = (-1, 0)
WGG_FN Rot2 cancel_y(Vec<2> v) { // rot2.wgsl:28-36: R with (R v).y == 0; the zero Rot2 if v.y is 0 already
    WGG_EXACT
    if (v.v[1] != 0.f) {
        const float r = w_sign(v.v[0]) / w_length2(v.v[0], v.v[1]);
        return Rot2{ v.v[0] * r, -v.v[1] * r };
    }
    return Rot2{ 0.f, 0.f };
}
WGG_FN Rot2 inv(Rot2 r) { return Rot2{ r.cos, -r.sin }; }
WGG_FN Rot2 mul(Rot2 a, Rot2 b) { // rot2.wgsl:61-65
    WGG_EXACT
    return Rot2{ a.cos * b.cos - a.sin * b.sin, a.sin * b.cos + a.cos * b.sin };
}
WGG_FN Vec<2> mulVec(Rot2 r, Vec<2> v) { // rot2.wgsl:68-70
    WGG_EXACT
    return Vec<2>{ { r.cos * v.v[0] - r.sin * v.v[1], r.sin * v.v[0] + r.cos * v.v[1] } };
}
WGG_FN Vec<2> invMulVec(Rot2 r, Vec<2> v) { // rot2.wgsl:73-75
    WGG_EXACT
    return Vec<2>{ { r.cos * v.v[0] + r.sin * v.v[1], -r.sin * v.v[0] + r.cos * v.v[1] } };
}
WGG_FN Mat2 toMatrix(Rot2 r) {
    Mat2 m;
    m.c[0][0] = r.cos; m.c[0][1] = r.sin;
    m.c[1][0] = -r.sin; m.c[1][1] = r.cos;
    return m;
}
// rotate_rows3 / rotate_rows4 (rot2.wgsl:78-95): (m[i][r], m[i + 1][r]) <- invMulVec(rot, .) for every r -- columns i and i + 1 of the column-major
// matrix, i.e. two "rows" of the eigenvector matrix nalgebra keeps transposed
template <int N>
WGG_FN void rotate_rows(Rot2 rot, Mat<N> &m, uint32_t i) {
    for (int r = 0; r < N; ++r) {
        const Vec<2> rv = invMulVec(rot, Vec<2>{ { m.c[i][r], m.c[i + 1][r] } });
        m.c[i][r] = rv.v[0];
        m.c[i + 1][r] = rv.v[1];
    }
}
WGG_FN void rotate_rows3(Rot2 rot, Mat3 &m, uint32_t i) { rotate_rows<3>(rot, m, i); }
WGG_FN void rotate_rows4(Rot2 rot, Mat4 &m, uint32_t i) { rotate_rows<4>(rot, m, i); }
} // namespace rot2

// ---------------------------------------------------------------------------------------------------------------
// symmetric eigendecomposition, M = V diag(lambda) V^T (eig2.wgsl:15-56, eig3.wgsl:24-282, eig4.wgsl:24-284)
//   N = 2: closed form; eigenvalues ((a + b + sigma) / 2, (a + b - sigma) / 2) in THAT order, eigenvectors normalised with last component +;
//          the identity basis and (a, b) as they come when the off-diagonal entry is exactly 0.
//   N = 3, 4: nalgebra's SymmetricEigen as the reference ported it -- scale by the largest |entry|, Householder tridiagonalisation, implicit
//          Wilkinson-shift QR steps on the unreduced block found by delimit_subproblem, the closed form for a 2 x 2 block; eigenvalues in the order
//          the deflation leaves them (unsorted). The reference's loop has no iteration cap; ITERATION_CAP below only bounds inputs (NaN-free inputs
//          never come near it: a 4 x 4 needs < 20 sweeps) on which the reference itself would not return.
// ---------------------------------------------------------------------------------------------------------------
template <int N>
struct SymmetricEigen {
    Mat<N> eigenvectors;
    Vec<N> eigenvalues;
};
namespace eig2 {
WGG_FN Vec<2> eigenvalues(const Mat2 &m) { // eig2.wgsl:44-56
    WGG_EXACT
    const float a = m.c[0][0], c = m.c[0][1], b = m.c[1][1];
    if (c == 0.f) return Vec<2>{ { a, b } };
    const float ab = a - b;
    const float sigma = sqrtf(4.f * c * c + ab * ab);
    return Vec<2>{ { (a + b + sigma) / 2.f, (a + b - sigma) / 2.f } };
}
WGG_FN SymmetricEigen<2> symmetric_eigen(const Mat2 &m) { // eig2.wgsl:15-42
    WGG_EXACT
    const float a = m.c[0][0], c = m.c[0][1], b = m.c[1][1];
    SymmetricEigen<2> r;
    if (c == 0.f) {
        r.eigenvectors = identity<2>();
        r.eigenvalues = Vec<2>{ { a, b } };
        return r;
    }
    const float ab = a - b;
    const float sigma = sqrtf(4.f * c * c + ab * ab);
    r.eigenvalues = Vec<2>{ { (a + b + sigma) / 2.f, (a + b - sigma) / 2.f } };
    const float e1 = (a - b + sigma) / (2.f * c), e2 = (a - b - sigma) / (2.f * c);
    const float l1 = w_length2(e1, 1.f), l2 = w_length2(e2, 1.f);
    r.eigenvectors.c[0][0] = e1 / l1; r.eigenvectors.c[0][1] = 1.f / l1;
    r.eigenvectors.c[1][0] = e2 / l2; r.eigenvectors.c[1][1] = 1.f / l2;
    return r;
}
} // namespace eig2

namespace eig_detail {
constexpr float EPS = 1.1920929e-7f; // eig3.wgsl:26
constexpr uint32_t ITERATION_CAP = 100000u;
template <int N>
struct Tridiag {
    Mat<N> m;
    float off_diag[N - 1];
};
template <int N>
WGG_FN Tridiag<N> tridiagonalize(const Mat<N> &x) { // eig3.wgsl:211-282: Householder axes stay in the lower triangle of m, one per column
    WGG_EXACT
    Tridiag<N> t;
    t.m = x;
    Mat<N> &m = t.m;
    for (int i = 0; i < N - 1; ++i) t.off_diag[i] = 0.f;
    for (int i = 0; i < N - 1; ++i) {
        float axis_sq_norm = 0.f; // the axis is m[i + 1.., i]
        for (int r = i + 1; r < N; ++r) axis_sq_norm = axis_sq_norm + m.c[i][r] * m.c[i][r];
        const float axis_norm = sqrtf(axis_sq_norm);
        const float modulus = fabsf(m.c[i][i + 1]);
        const float signed_norm = w_sign(m.c[i][i + 1]) * axis_norm;
        const float factor = (axis_sq_norm + modulus * axis_norm) * 2.f;
        m.c[i][i + 1] = m.c[i][i + 1] + signed_norm;
        if (factor != 0.f) {
            const float factor_sqrt = sqrtf(factor);
            float norm = 0.f;
            for (int r = i + 1; r < N; ++r) {
                m.c[i][r] = m.c[i][r] / factor_sqrt;
                norm = norm + m.c[i][r] * m.c[i][r];
            }
            norm = sqrtf(norm);
            for (int r = i + 1; r < N; ++r) m.c[i][r] = m.c[i][r] / norm;
            t.off_diag[i] = -signed_norm;
            float p[N]; // p = 2 M axis   (hegemv)
            for (int r = 0; r < N; ++r) p[r] = 0.f;
            for (int c = i + 1; c < N; ++c)
                for (int r = i + 1; r < N; ++r) p[r] = p[r] + 2.f * m.c[c][r] * m.c[i][c];
            float dot = 0.f;
            for (int r = i + 1; r < N; ++r) dot = dot + m.c[i][r] * p[r];
            for (int c = i + 1; c < N; ++c) // M <- M - p axis^T - axis p^T + 2 dot axis axis^T
                for (int r = i + 1; r < N; ++r)
                    m.c[c][r] = m.c[c][r] + (2.f * dot * m.c[i][r] * m.c[i][c] - p[r] * m.c[i][c] - m.c[i][r] * p[c]);
        } else {
            t.off_diag[i] = signed_norm;
        }
    }
    return t;
}
template <int N>
WGG_FN void delimit_subproblem(const float *diag, float *off_diag, uint32_t end, float eps, uint32_t &start_out, uint32_t &end_out) { // eig3.wgsl:162-198
    WGG_EXACT
    uint32_t n = end;
    while (n > 0u) {
        const uint32_t m = n - 1u;
        if (fabsf(off_diag[m]) > eps * (fabsf(diag[n]) + fabsf(diag[m]))) break;
        n -= 1u;
    }
    if (n == 0u) { start_out = 0u; end_out = 0u; return; }
    uint32_t new_start = n - 1u;
    while (new_start > 0u) {
        const uint32_t m = new_start - 1u;
        if (off_diag[m] == 0.f || fabsf(off_diag[m]) <= eps * (fabsf(diag[new_start]) + fabsf(diag[m]))) {
            off_diag[m] = 0.f;
            break;
        }
        new_start -= 1u;
    }
    start_out = new_start;
    end_out = n;
}
WGG_FN float wilkinson_shift(float tmm, float tnn, float tmn) { // eig3.wgsl:200-209
    WGG_EXACT
    const float sq_tmn = tmn * tmn;
    if (sq_tmn != 0.f) {
        const float d = (tmm - tnn) * 0.5f;
        return tnn - sq_tmn / (d + w_sign(d) * sqrtf(d * d + sq_tmn));
    }
    return tnn;
}
} // namespace eig_detail

template <int N>
WGG_FN SymmetricEigen<N> symmetric_eigen(const Mat<N> &x) { // eig3.wgsl:24-160 (N = 3), eig4.wgsl:24-162 (N = 4)
    WGG_EXACT
    using namespace eig_detail;
    Mat<N> m = x;
    const float m_amax = min_max::amaxm<N>(x);
    if (m_amax != 0.f)
        for (int c = 0; c < N; ++c)
            for (int r = 0; r < N; ++r) m.c[c][r] = m.c[c][r] / m_amax;
    const Tridiag<N> tri = tridiagonalize<N>(m);
    float diag[N], off_diag[N - 1];
    for (int k = 0; k < N; ++k) diag[k] = tri.m.c[k][k];
    for (int k = 0; k < N - 1; ++k) off_diag[k] = fabsf(tri.off_diag[k]); // SymmetricTridiagonal::unpack takes the modulus
    Mat<N> q = identity<N>(); // householder::assemble_q
    for (int i = N - 2; i >= 0; --i) {
        const float sgn = w_sign(tri.off_diag[i]);
        for (int c = i; c < N; ++c) {
            const float m_two = -2.f * sgn;
            float f = 0.f;
            for (int r = i + 1; r < N; ++r) f = f + tri.m.c[i][r] * q.c[c][r];
            for (int r = i + 1; r < N; ++r) q.c[c][r] = m_two * f * tri.m.c[i][r] + q.c[c][r] * sgn;
        }
    }
    uint32_t start, end;
    delimit_subproblem<N>(diag, off_diag, (uint32_t)N - 1u, EPS, start, end);
    uint32_t niter = 0;
    while (end != start && niter < ITERATION_CAP) {
        const uint32_t subdim = end - start + 1u;
        if (subdim > 2u) {
            const uint32_t mm = end - 1u, n = end;
            const float shift = wilkinson_shift(diag[mm], diag[n], off_diag[mm]);
            Vec<2> v{ { diag[start] - shift, off_diag[start] } };
            for (uint32_t i = start; i < n; ++i) {
                const uint32_t j = i + 1u;
                const Rot2 rot = rot2::cancel_y(v);
                if (!rot2::is_valid(rot)) break;
                if (i > start) off_diag[i - 1] = w_sign(v.v[0]) * w_length2(v.v[0], v.v[1]);
                const float mii = diag[i], mjj = diag[j], mij = off_diag[i];
                const float cc = rot.cos * rot.cos, ss = rot.sin * rot.sin, cs = rot.cos * rot.sin;
                const float b = cs * 2.f * mij;
                diag[i] = (cc * mii + ss * mjj) - b;
                diag[j] = (ss * mii + cc * mjj) + b;
                off_diag[i] = cs * (mii - mjj) + mij * (cc - ss);
                if (i != n - 1u) {
                    v.v[0] = off_diag[i];
                    v.v[1] = -rot.sin * off_diag[i + 1];
                    off_diag[i + 1] = off_diag[i + 1] * rot.cos;
                }
                rot2::rotate_rows<N>(rot2::inv(rot), q, i);
            }
            if (fabsf(off_diag[mm]) <= EPS * (fabsf(diag[mm]) + fabsf(diag[n]))) end -= 1u;
        } else if (subdim == 2u) {
            Mat2 m2;
            m2.c[0][0] = diag[start]; m2.c[0][1] = off_diag[start];
            m2.c[1][0] = off_diag[start]; m2.c[1][1] = diag[start + 1];
            const Vec<2> eigvals = eig2::eigenvalues(m2);
            const float bx = eigvals.v[0] - diag[start + 1], by = off_diag[start];
            diag[start] = eigvals.v[0];
            diag[start + 1] = eigvals.v[1];
            const float basis_len = w_length2(bx, by);
            if (basis_len > EPS) {
                const float s = w_sign(bx) / basis_len;
                rot2::rotate_rows<N>(Rot2{ bx * s, by * s }, q, start);
            }
            end -= 1u;
        }
        delimit_subproblem<N>(diag, off_diag, end, EPS, start, end); // decoupling may have happened
        ++niter;
    }
    SymmetricEigen<N> r;
    r.eigenvectors = q;
    for (int k = 0; k < N; ++k) r.eigenvalues.v[k] = diag[k] * m_amax;
    return r;
}
template <>
WGG_FN SymmetricEigen<2> symmetric_eigen<2>(const Mat2 &x) { return eig2::symmetric_eigen(x); }

// ---------------------------------------------------------------------------------------------------------------
// Quat (quat.wgsl): unit quaternion, coords = (x, y, z, w)
// ---------------------------------------------------------------------------------------------------------------
struct Quat {
    float x, y, z, w;
};
namespace quat {
// WGSL builtins as their defining formulas, left to right (what the tests' WGSL executor evaluates)
WGG_FN float dot3(const float *a, const float *b) { WGG_EXACT return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
WGG_FN Vec<3> cross3(const float *a, const float *b) {
    WGG_EXACT
    return Vec<3>{ { a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1] } };
}
WGG_FN Quat identity() { return Quat{ 0.f, 0.f, 0.f, 1.f }; }
WGG_FN Quat fromScaledAxis(Vec<3> aa) { // quat.wgsl:16-28: rotation of |aa| radians about aa / |aa|
    WGG_EXACT
    const float angle = sqrtf(dot3(aa.v, aa.v));
    if (angle == 0.f) return identity();
    const float hs = w_sin(angle / 2.f), hc = w_cos(angle / 2.f);
    return Quat{ (aa.v[0] / angle) * hs, (aa.v[1] / angle) * hs, (aa.v[2] / angle) * hs, hc };
}
WGG_FN Vec<3> imag(Quat q) { return Vec<3>{ { q.x, q.y, q.z } }; }
WGG_FN Quat inv(Quat q) { return Quat{ -q.x, -q.y, -q.z, q.w }; } // conjugate (unit quaternion)
WGG_FN Quat mul(Quat l, Quat r) { // quat.wgsl:73-77
    WGG_EXACT
    const float lv[3] = { l.x, l.y, l.z }, rv[3] = { r.x, r.y, r.z };
    const float scalar = l.w * r.w - dot3(lv, rv);
    const Vec<3> c = cross3(lv, rv);
    return Quat{ c.v[0] + l.w * rv[0] + r.w * lv[0], c.v[1] + l.w * rv[1] + r.w * lv[1], c.v[2] + l.w * rv[2] + r.w * lv[2], scalar };
}
WGG_FN Quat renormalizeFast(Quat q) { // quat.wgsl:59-62: one Newton step of 1/sqrt(|q|^2) around 1
    WGG_EXACT
    const float sq_norm = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    const float f = 0.5f * (3.f - sq_norm);
    return Quat{ q.x * f, q.y * f, q.z * f, q.w * f };
}
WGG_FN Vec<3> mulVec(Quat q, Vec<3> v) { // quat.wgsl:80-84: t = 2 (u x v); t w + u x t + v
    WGG_EXACT
    const float u[3] = { q.x, q.y, q.z };
    Vec<3> t = cross3(u, v.v);
    for (int i = 0; i < 3; ++i) t.v[i] = t.v[i] * 2.f;
    const Vec<3> c = cross3(u, t.v);
    return Vec<3>{ { t.v[0] * q.w + c.v[0] + v.v[0], t.v[1] * q.w + c.v[1] + v.v[1], t.v[2] * q.w + c.v[2] + v.v[2] } };
}
WGG_FN Vec<3> invMulVec(Quat q, Vec<3> v) { // quat.wgsl:87-91: the same with -w
    WGG_EXACT
    const float u[3] = { q.x, q.y, q.z };
    Vec<3> t = cross3(u, v.v);
    for (int i = 0; i < 3; ++i) t.v[i] = t.v[i] * 2.f;
    const Vec<3> c = cross3(u, t.v);
    const float nw = -q.w;
    return Vec<3>{ { t.v[0] * nw + c.v[0] + v.v[0], t.v[1] * nw + c.v[1] + v.v[1], t.v[2] * nw + c.v[2] + v.v[2] } };
}
WGG_FN Mat3 toMatrix(Quat q) { // quat.wgsl:31-54 -- ww + ii - jj - kk on the diagonal (NOT 1 - 2 (jj + kk): the two differ for the drifted,
    WGG_EXACT                  // non-unit quaternions renormalizeFast exists for)
    const float i = q.x, j = q.y, k = q.z, w = q.w;
    const float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    const float ij = i * j * 2.f, wk = w * k * 2.f, wj = w * j * 2.f, ik = i * k * 2.f, jk = j * k * 2.f, wi = w * i * 2.f;
    Mat3 m;
    m.c[0][0] = ww + ii - jj - kk; m.c[0][1] = wk + ij;           m.c[0][2] = ik - wj;
    m.c[1][0] = ij - wk;           m.c[1][1] = ww - ii + jj - kk; m.c[1][2] = wi + jk;
    m.c[2][0] = wj + ik;           m.c[2][1] = jk - wi;           m.c[2][2] = ww - ii - jj + kk;
    return m;
}
} // namespace quat

// ---------------------------------------------------------------------------------------------------------------
// SVD, M = U diag(S) Vt (svd2.wgsl:12-46, svd3.wgsl:56-312; the reference has no svd4)
//   N = 2: closed form through two stable_atan2 angles; S = (q + r, |q - r|), the sign of q - r moved into Vt's rows.
//   N = 3: McAdams et al., "Computing the SVD of 3x3 matrices with minimal branching and elementary floating point operations", as the reference
//          ported it from tbtSVD: JACOBI_STEPS fixed sweeps of approximate Givens conjugations on A^T A accumulated in a quaternion, columns of
//          B = A V sorted by norm with sign-flipping swaps, then a Givens-quaternion QR of B: U = Q, S = diag(R) (the last one may be negative:
//          the reference leaves it so). Only + - * / and fma occur, and the reciprocal square roots are the bit-trick + Newton steps of the
//          reference ("allows for exact matching results on CPU and GPU", svd3.wgsl:55): every build of this header returns the same bits.
// ---------------------------------------------------------------------------------------------------------------
template <int N>
struct Svd {
    Mat<N> u;
    Vec<N> s;
    Mat<N> vt;
};
namespace svd2 {
WGG_FN Svd<2> svd(const Mat2 &m) { // svd2.wgsl:12-40
    WGG_EXACT
    const float e = (m.c[0][0] + m.c[1][1]) * 0.5f;
    const float f = (m.c[0][0] - m.c[1][1]) * 0.5f;
    const float g = (m.c[0][1] + m.c[1][0]) * 0.5f;
    const float h = (m.c[0][1] - m.c[1][0]) * 0.5f;
    const float q = sqrtf(e * e + h * h);
    const float r = sqrtf(f * f + g * g);
    const float sx = q + r, sy = q - r;
    const float sy_sign = sy < 0.f ? -1.f : 1.f;
    const float a1 = trig::stable_atan2(g, f);
    const float a2 = trig::stable_atan2(h, e);
    const float theta = (a2 - a1) * 0.5f;
    const float phi = (a2 + a1) * 0.5f;
    const float st = w_sin(theta), ct = w_cos(theta), sp = w_sin(phi), cp = w_cos(phi);
    Svd<2> o;
    o.s = Vec<2>{ { sx, sy * sy_sign } };
    o.u.c[0][0] = cp; o.u.c[0][1] = sp;
    o.u.c[1][0] = -sp; o.u.c[1][1] = cp;
    o.vt.c[0][0] = ct; o.vt.c[0][1] = st * sy_sign;
    o.vt.c[1][0] = -st; o.vt.c[1][1] = ct * sy_sign;
    return o;
}
} // namespace svd2

namespace svd3 {
constexpr float GAMMA = 5.828427124f;  // sqrt(8) + 3   (svd3.wgsl:19-27)
constexpr float CSTAR = 0.923879532f;  // cos(pi / 8)
constexpr float SSTAR = 0.3826834323f; // sin(pi / 8)
constexpr float SVD_EPSILON = 1e-6f;
constexpr uint32_t JACOBI_STEPS = 12, RSQRT_STEPS = 4, RSQRT1_STEPS = 6;
struct Symmetric3x3 {
    float mxx, myx, myy, mzx, mzy, mzz;
};
struct givens {
    float ch, sh;
};
WGG_FN float rsqrt_steps(float val, uint32_t steps) { // svd3.wgsl:56-80: the 0x5f375a82 seed, then `steps` Newton iterations x <- x fma(x x, -val / 2, 1.5)
    WGG_EXACT
    float x = val;
    const float xhalf = -0.5f * x;
    int32_t i;
    __builtin_memcpy(&i, &x, 4);
    i = 0x5f375a82 - (i >> 1);
    __builtin_memcpy(&x, &i, 4);
    for (uint32_t k = 0; k < steps; ++k) x = x * __builtin_fmaf(x * x, xhalf, 1.5f);
    return x;
}
WGG_FN float rsqrt(float val) { return rsqrt_steps(val, RSQRT_STEPS); }
WGG_FN float rsqrt1(float val) { return rsqrt_steps(val, RSQRT1_STEPS); }
WGG_FN float accurateSqrt(float x) { return 1.f / rsqrt1(x); } // svd3.wgsl:83-85
WGG_FN void condSwap(bool c, float &x, float &y) { const float x0 = x; x = c ? y : x; y = c ? x0 : y; }        // svd3.wgsl:88-92
WGG_FN void condNegSwap(bool c, float &x, float &y) { const float x0 = -x; x = c ? y : x; y = c ? x0 : y; }    // svd3.wgsl:95-99
WGG_FN void condNegSwapVec(bool c, float *x, float *y) {                                                       // svd3.wgsl:107-111
    for (int k = 0; k < 3; ++k) condNegSwap(c, x[k], y[k]);
}
WGG_FN givens approximateGivensQuaternion(const Symmetric3x3 &A) { // svd3.wgsl:116-129 (Algorithm 2)
    WGG_EXACT
    const givens g{ 2.f * (A.mxx - A.myy), A.myx };
    bool b = GAMMA * g.sh * g.sh < g.ch * g.ch;
    const float w = rsqrt(__builtin_fmaf(g.ch, g.ch, g.sh * g.sh));
    if (w != w) b = false;
    return b ? givens{ w * g.ch, w * g.sh } : givens{ CSTAR, SSTAR };
}
WGG_FN void jacobiConjugation(int x, int y, int z, Symmetric3x3 &S, float *q) { // svd3.wgsl:132-166; q = quaternion coords (x, y, z, w)
    WGG_EXACT
    givens g = approximateGivensQuaternion(S);
    const float scale = 1.f / __builtin_fmaf(g.ch, g.ch, g.sh * g.sh);
    const float a = __builtin_fmaf(g.ch, g.ch, -g.sh * g.sh) * scale;
    const float b = 2.f * g.sh * g.ch * scale;
    Symmetric3x3 _S = S;
    // S = Q' S Q
    S.mxx = __builtin_fmaf(a, __builtin_fmaf(a, _S.mxx, b * _S.myx), b * (__builtin_fmaf(a, _S.myx, b * _S.myy)));
    S.myx = __builtin_fmaf(a, __builtin_fmaf(-b, _S.mxx, a * _S.myx), b * (__builtin_fmaf(-b, _S.myx, a * _S.myy)));
    S.myy = __builtin_fmaf(-b, __builtin_fmaf(-b, _S.mxx, a * _S.myx), a * (__builtin_fmaf(-b, _S.myx, a * _S.myy)));
    S.mzx = __builtin_fmaf(a, _S.mzx, b * _S.mzy);
    S.mzy = __builtin_fmaf(-b, _S.mzx, a * _S.mzy);
    S.mzz = _S.mzz;
    // the cumulative rotation
    const float tmp[3] = { q[0] * g.sh, q[1] * g.sh, q[2] * g.sh };
    g.sh = g.sh * q[3];
    // (x, y, z) is ((0, 1, 2), (1, 2, 0), (2, 0, 1)) for (p, q) = ((0, 1), (1, 2), (0, 2))
    q[z] = __builtin_fmaf(q[z], g.ch, g.sh);
    q[3] = __builtin_fmaf(q[3], g.ch, -tmp[z]);
    q[x] = __builtin_fmaf(q[x], g.ch, tmp[y]);
    q[y] = __builtin_fmaf(q[y], g.ch, -tmp[x]);
    // re-arrange the matrix for the next pair
    _S.mxx = S.myy;
    _S.myx = S.mzy; _S.myy = S.mzz;
    _S.mzx = S.myx; _S.mzy = S.mzx; _S.mzz = S.mxx;
    S = _S;
}
WGG_FN Quat jacobiEigenanalysis(const Symmetric3x3 &S) { // svd3.wgsl:170-179
    Symmetric3x3 mat = S;
    float q[4] = { 0.f, 0.f, 0.f, 1.f };
    for (uint32_t i = 0; i < JACOBI_STEPS; ++i) {
        jacobiConjugation(0, 1, 2, mat, q);
        jacobiConjugation(1, 2, 0, mat, q);
        jacobiConjugation(2, 0, 1, mat, q);
    }
    return Quat{ q[0], q[1], q[2], q[3] };
}
WGG_FN void sortSingularValues(Mat3 &B, Mat3 &V) { // svd3.wgsl:190-214 (Algorithm 3): columns by decreasing norm, a swap negates one column
    WGG_EXACT
    float rho1 = quat::dot3(B.c[0], B.c[0]), rho2 = quat::dot3(B.c[1], B.c[1]), rho3 = quat::dot3(B.c[2], B.c[2]);
    bool c = rho1 < rho2;
    condNegSwapVec(c, B.c[0], B.c[1]);
    condNegSwapVec(c, V.c[0], V.c[1]);
    condSwap(c, rho1, rho2);
    c = rho1 < rho3;
    condNegSwapVec(c, B.c[0], B.c[2]);
    condNegSwapVec(c, V.c[0], V.c[2]);
    condSwap(c, rho1, rho3);
    c = rho2 < rho3;
    condNegSwapVec(c, B.c[1], B.c[2]);
    condNegSwapVec(c, V.c[1], V.c[2]);
}
WGG_FN givens QRGivensQuaternion(float a1, float a2) { // svd3.wgsl:217-230 (Algorithm 4): a1 = the pivot, a2 = the entry to annihilate
    WGG_EXACT
    const float epsilon = SVD_EPSILON;
    const float rho = accurateSqrt(__builtin_fmaf(a1, a1, a2 * a2));
    float ch = fabsf(a1) + w_max(rho, epsilon);
    float sh = rho > epsilon ? a2 : 0.f;
    const bool b = a1 < 0.f;
    condSwap(b, sh, ch);
    const float w = rsqrt(__builtin_fmaf(ch, ch, sh * sh));
    ch = ch * w;
    sh = sh * w;
    return givens{ ch, sh };
}
struct QR3 {
    Mat3 Q, R;
};
WGG_FN QR3 QRDecomposition(const Mat3 &B) { // svd3.wgsl:233-293 (section 4.2): three Givens quaternions, Q = Q1 Q2 Q3 in closed form
    WGG_EXACT
    // first rotation (ch, 0, 0, sh)
    const givens g1 = QRGivensQuaternion(B.c[0][0], B.c[0][1]);
    float a = __builtin_fmaf(-2.f, g1.sh * g1.sh, 1.f);
    float b = 2.f * g1.ch * g1.sh;
    float r00 = __builtin_fmaf(a, B.c[0][0], b * B.c[0][1]), r01 = __builtin_fmaf(a, B.c[1][0], b * B.c[1][1]), r02 = __builtin_fmaf(a, B.c[2][0], b * B.c[2][1]);
    float r10 = __builtin_fmaf(-b, B.c[0][0], a * B.c[0][1]), r11 = __builtin_fmaf(-b, B.c[1][0], a * B.c[1][1]), r12 = __builtin_fmaf(-b, B.c[2][0], a * B.c[2][1]);
    float r20 = B.c[0][2], r21 = B.c[1][2], r22 = B.c[2][2];
    // second rotation (ch, 0, -sh, 0)
    const givens g2 = QRGivensQuaternion(r00, r20);
    a = __builtin_fmaf(-2.f, g2.sh * g2.sh, 1.f);
    b = 2.f * g2.ch * g2.sh;
    const float b00 = __builtin_fmaf(a, r00, b * r20), b01 = __builtin_fmaf(a, r01, b * r21), b02 = __builtin_fmaf(a, r02, b * r22);
    const float b10 = r10, b11 = r11, b12 = r12;
    const float b20 = __builtin_fmaf(-b, r00, a * r20), b21 = __builtin_fmaf(-b, r01, a * r21), b22 = __builtin_fmaf(-b, r02, a * r22);
    // third rotation (ch, sh, 0, 0)
    const givens g3 = QRGivensQuaternion(b11, b21);
    a = __builtin_fmaf(-2.f, g3.sh * g3.sh, 1.f);
    b = 2.f * g3.ch * g3.sh;
    r00 = b00; r01 = b01; r02 = b02;
    r10 = __builtin_fmaf(a, b10, b * b20); r11 = __builtin_fmaf(a, b11, b * b21); r12 = __builtin_fmaf(a, b12, b * b22);
    r20 = __builtin_fmaf(-b, b10, a * b20); r21 = __builtin_fmaf(-b, b11, a * b21); r22 = __builtin_fmaf(-b, b12, a * b22);
    // Q = Q1 Q2 Q3
    const float sh12 = 2.f * __builtin_fmaf(g1.sh, g1.sh, -0.5f);
    const float sh22 = 2.f * __builtin_fmaf(g2.sh, g2.sh, -0.5f);
    const float sh32 = 2.f * __builtin_fmaf(g3.sh, g3.sh, -0.5f);
    const float q00 = sh12 * sh22;
    const float q01 = __builtin_fmaf(4.f * g2.ch * g3.ch, sh12 * g2.sh * g3.sh, 2.f * g1.ch * g1.sh * sh32);
    const float q02 = __builtin_fmaf(4.f * g1.ch * g3.ch, g1.sh * g3.sh, -2.f * g2.ch * sh12 * g2.sh * sh32);
    const float q10 = -2.f * g1.ch * g1.sh * sh22;
    const float q11 = __builtin_fmaf(-8.f * g1.ch * g2.ch * g3.ch, g1.sh * g2.sh * g3.sh, sh12 * sh32);
    const float q12 = __builtin_fmaf(-2.f * g3.ch, g3.sh, 4.f * g1.sh * __builtin_fmaf(g3.ch * g1.sh, g3.sh, g1.ch * g2.ch * g2.sh * sh32));
    const float q20 = 2.f * g2.ch * g2.sh;
    const float q21 = -2.f * g3.ch * sh22 * g3.sh;
    const float q22 = sh22 * sh32;
    QR3 o;
    o.Q.c[0][0] = q00; o.Q.c[0][1] = q10; o.Q.c[0][2] = q20;
    o.Q.c[1][0] = q01; o.Q.c[1][1] = q11; o.Q.c[1][2] = q21;
    o.Q.c[2][0] = q02; o.Q.c[2][1] = q12; o.Q.c[2][2] = q22;
    o.R.c[0][0] = r00; o.R.c[0][1] = r10; o.R.c[0][2] = r20;
    o.R.c[1][0] = r01; o.R.c[1][1] = r11; o.R.c[1][2] = r21;
    o.R.c[2][0] = r02; o.R.c[2][1] = r12; o.R.c[2][2] = r22;
    return o;
}
WGG_FN Svd<3> svd(const Mat3 &A) { // svd3.wgsl:296-307
    const Mat3 ata = mul_exact<3>(transpose<3>(A), A);
    const Symmetric3x3 ata_sym{ ata.c[0][0], ata.c[0][1], ata.c[1][1], ata.c[0][2], ata.c[1][2], ata.c[2][2] };
    Mat3 V = quat::toMatrix(jacobiEigenanalysis(ata_sym));
    Mat3 B = mul_exact<3>(A, V);
    sortSingularValues(B, V);
    const QR3 qr = QRDecomposition(B);
    Svd<3> o;
    o.u = qr.Q;
    o.s = Vec<3>{ { qr.R.c[0][0], qr.R.c[1][1], qr.R.c[2][2] } };
    o.vt = transpose<3>(V);
    return o;
}
} // namespace svd3

template <int N>
WGG_FN Svd<N> svd(const Mat<N> &x);
template <>
WGG_FN Svd<2> svd<2>(const Mat2 &x) { return svd2::svd(x); }
template <>
WGG_FN Svd<3> svd<3>(const Mat3 &x) { return svd3::svd(x); }
template <int N>
WGG_FN Mat<N> recompose(const Svd<N> &d) { // svd2.wgsl:43-46, svd3.wgsl:309-312: (U with column j scaled by S[j]) Vt
    WGG_EXACT
    Mat<N> us = d.u;
    for (int j = 0; j < N; ++j)
        for (int k = 0; k < N; ++k) us.c[j][k] = us.c[j][k] * d.s.v[j];
    return mul_exact<N>(us, d.vt);
}

// ---------------------------------------------------------------------------------------------------------------
// Sim2 / Sim3 (sim2.wgsl, sim3.wgsl): x -> scale * R x + translation
// ---------------------------------------------------------------------------------------------------------------
struct Sim2 {
    Rot2 rotation;
    Vec<2> translation;
    float scale;
};
namespace sim2 {
WGG_FN Sim2 identity() { return Sim2{ rot2::identity(), Vec<2>{ { 0.f, 0.f } }, 1.f }; }
WGG_FN Vec<2> mulVec(const Sim2 &s, Vec<2> v) { WGG_EXACT Vec<2> r = rot2::mulVec(s.rotation, v); r.v[0] = r.v[0] * s.scale; r.v[1] = r.v[1] * s.scale; return r; } // sim2.wgsl:57-59
WGG_FN Vec<2> mulUnitVec(const Sim2 &s, Vec<2> v) { return rot2::mulVec(s.rotation, v); }
WGG_FN Vec<2> mulPt(const Sim2 &s, Vec<2> p) { // sim2.wgsl:42-44: R (p scale) + t
    WGG_EXACT
    const Vec<2> r = rot2::mulVec(s.rotation, Vec<2>{ { p.v[0] * s.scale, p.v[1] * s.scale } });
    return Vec<2>{ { r.v[0] + s.translation.v[0], r.v[1] + s.translation.v[1] } };
}
WGG_FN Vec<2> invMulVec(const Sim2 &s, Vec<2> v) { WGG_EXACT Vec<2> r = rot2::invMulVec(s.rotation, v); r.v[0] = r.v[0] / s.scale; r.v[1] = r.v[1] / s.scale; return r; }
WGG_FN Vec<2> invMulUnitVec(const Sim2 &s, Vec<2> v) { return rot2::invMulVec(s.rotation, v); }
WGG_FN Vec<2> invMulPt(const Sim2 &s, Vec<2> p) { // sim2.wgsl:47-49
    WGG_EXACT
    return invMulVec(s, Vec<2>{ { p.v[0] - s.translation.v[0], p.v[1] - s.translation.v[1] } });
}
WGG_FN Sim2 mul(const Sim2 &a, const Sim2 &b) { // sim2.wgsl:21-25: t_a + (R_a t_b) scale_a
    WGG_EXACT
    const Vec<2> r = rot2::mulVec(a.rotation, b.translation);
    return Sim2{ rot2::mul(a.rotation, b.rotation), Vec<2>{ { a.translation.v[0] + r.v[0] * a.scale, a.translation.v[1] + r.v[1] * a.scale } }, a.scale * b.scale };
}
WGG_FN Sim2 inv(const Sim2 &s) { // sim2.wgsl:34-39: R^-1 (-t) times the RECIPROCAL scale
    WGG_EXACT
    const float scale = 1.f / s.scale;
    const Rot2 rotation = rot2::inv(s.rotation);
    const Vec<2> t = rot2::mulVec(rotation, Vec<2>{ { -s.translation.v[0], -s.translation.v[1] } });
    return Sim2{ rotation, Vec<2>{ { t.v[0] * scale, t.v[1] * scale } }, scale };
}
WGG_FN Sim2 invMul(const Sim2 &a, const Sim2 &b) { return mul(inv(a), b); }
} // namespace sim2

struct Sim3 {
    Quat rotation;
    Vec<3> translation;
    float scale; // the reference packs (translation, scale) into one vec4: translation_scale
};
namespace sim3 {
WGG_FN Sim3 identity() { return Sim3{ quat::identity(), Vec<3>{ { 0.f, 0.f, 0.f } }, 1.f }; }
WGG_FN Vec<3> mulVec(const Sim3 &s, Vec<3> v) { WGG_EXACT Vec<3> r = quat::mulVec(s.rotation, v); for (int i = 0; i < 3; ++i) r.v[i] = r.v[i] * s.scale; return r; }
WGG_FN Vec<3> mulUnitVec(const Sim3 &s, Vec<3> v) { return quat::mulVec(s.rotation, v); }
WGG_FN Vec<3> mulPt(const Sim3 &s, Vec<3> p) { // sim3.wgsl:40-42: R (p scale) + t
    WGG_EXACT
    const Vec<3> r = quat::mulVec(s.rotation, Vec<3>{ { p.v[0] * s.scale, p.v[1] * s.scale, p.v[2] * s.scale } });
    return Vec<3>{ { r.v[0] + s.translation.v[0], r.v[1] + s.translation.v[1], r.v[2] + s.translation.v[2] } };
}
WGG_FN Vec<3> invMulVec(const Sim3 &s, Vec<3> v) { WGG_EXACT Vec<3> r = quat::invMulVec(s.rotation, v); for (int i = 0; i < 3; ++i) r.v[i] = r.v[i] / s.scale; return r; }
WGG_FN Vec<3> invMulUnitVec(const Sim3 &s, Vec<3> v) { return quat::invMulVec(s.rotation, v); }
WGG_FN Vec<3> invMulPt(const Sim3 &s, Vec<3> p) { // sim3.wgsl:45-47
    WGG_EXACT
    return invMulVec(s, Vec<3>{ { p.v[0] - s.translation.v[0], p.v[1] - s.translation.v[1], p.v[2] - s.translation.v[2] } });
}
WGG_FN Sim3 mul(const Sim3 &a, const Sim3 &b) { // sim3.wgsl:19-23
    WGG_EXACT
    const Vec<3> r = quat::mulVec(a.rotation, b.translation);
    return Sim3{ quat::mul(a.rotation, b.rotation),
                 Vec<3>{ { a.translation.v[0] + r.v[0] * a.scale, a.translation.v[1] + r.v[1] * a.scale, a.translation.v[2] + r.v[2] * a.scale } }, a.scale * b.scale };
}
WGG_FN Sim3 inv(const Sim3 &s) { // sim3.wgsl:26-31
    WGG_EXACT
    const float scale = 1.f / s.scale;
    const Quat rotation = quat::inv(s.rotation);
    const Vec<3> t = quat::mulVec(rotation, Vec<3>{ { -s.translation.v[0], -s.translation.v[1], -s.translation.v[2] } });
    return Sim3{ rotation, Vec<3>{ { t.v[0] * scale, t.v[1] * scale, t.v[2] * scale } }, scale };
}
WGG_FN Sim3 invMul(const Sim3 &a, const Sim3 &b) { return mul(inv(a), b); }
} // namespace sim3

} // namespace geometry
} // namespace wgebra
