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
//   * qr                     (qr2/3/4.wgsl)   : Householder; Q orthonormal, R upper triangular with a NON-NEGATIVE diagonal
//                                               (nalgebra's convention, which makes the factorisation unique for full rank);
//   * symmetric_eigen        (eig2/3/4.wgsl)  : M = V diag(lambda) V^T for symmetric M; eigenvalue order unspecified (the
//                                               reference's tests check the reconstruction); here: cyclic Jacobi rotations;
//   * svd (2, 3)             (svd2/3.wgsl)    : M = U diag(S) Vt, S >= 0 in descending order;
//   * Quat, Rot2, Sim2, Sim3 (quat/rot2/sim2/sim3.wgsl): unit quaternion (x, y, z, w), 2-D rotation (cos, sin), similarities
//                                               x -> scale * R x + t, with the reference's function names.
// Results agree with the reference within its own test tolerances (relative 1e-3 / 1e-4 against nalgebra). The CLOSED-FORM functions -- inv2/3/4,
// cholesky, lu, Rot2, Quat, Sim2, Sim3 -- follow the reference's WGSL expression by expression, every product and sum rounded separately (no FMA
// contraction: WGG_EXACT below), so that on the same inputs they give the bits the WGSL text gives when read that way (left to right, `dot` and
// `cross` as their defining formulas: tests/golden/wgsl_exec_geometry.npz, made by running the reference's .wgsl files through the tests' WGSL executor).
// QR, eigen and SVD are different but equivalent algorithms and are not bit-identical.
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
// QR (qr2/3/4.wgsl): Householder reflections, then signs fixed so that diag(R) >= 0
// ---------------------------------------------------------------------------------------------------------------
template <int N>
struct QR {
    Mat<N> q, r;
};
template <int N>
WGG_FN QR<N> qr(const Mat<N> &x) {
    QR<N> o;
    o.r = x;
    o.q = identity<N>();
    for (int i = 0; i < N; ++i) {
        float n2 = 0.f;
        for (int rr = i; rr < N; ++rr) n2 += o.r.c[i][rr] * o.r.c[i][rr];
        const float nrm = sqrtf(n2);
        if (nrm == 0.f) continue;
        float v[N];
        for (int rr = 0; rr < N; ++rr) v[rr] = rr >= i ? o.r.c[i][rr] : 0.f;
        v[i] += (v[i] >= 0.f ? nrm : -nrm);
        float vn2 = 0.f;
        for (int rr = i; rr < N; ++rr) vn2 += v[rr] * v[rr];
        if (vn2 == 0.f) continue;
        const float beta = 2.f / vn2;
        // R <- H R,  Q <- Q H   (H = I - beta v v^T)
        for (int cc = 0; cc < N; ++cc) {
            float d = 0.f;
            for (int rr = i; rr < N; ++rr) d += v[rr] * o.r.c[cc][rr];
            d *= beta;
            for (int rr = i; rr < N; ++rr) o.r.c[cc][rr] -= d * v[rr];
        }
        for (int rr = 0; rr < N; ++rr) {
            float d = 0.f;
            for (int cc = i; cc < N; ++cc) d += o.q.c[cc][rr] * v[cc];
            d *= beta;
            for (int cc = i; cc < N; ++cc) o.q.c[cc][rr] -= d * v[cc];
        }
    }
    for (int i = 0; i < N; ++i) {
        for (int rr = i + 1; rr < N; ++rr) o.r.c[i][rr] = 0.f; // exact zeros below the diagonal
        if (o.r.c[i][i] < 0.f) {
            for (int cc = 0; cc < N; ++cc) o.r.c[cc][i] = -o.r.c[cc][i]; // row i of R
            for (int rr = 0; rr < N; ++rr) o.q.c[i][rr] = -o.q.c[i][rr]; // column i of Q
        }
    }
    return o;
}

// ---------------------------------------------------------------------------------------------------------------
// symmetric eigendecomposition (eig2/3/4.wgsl): cyclic Jacobi; M = V diag(lambda) V^T
// ---------------------------------------------------------------------------------------------------------------
template <int N>
struct SymmetricEigen {
    Mat<N> eigenvectors;
    Vec<N> eigenvalues;
};
template <int N>
WGG_FN SymmetricEigen<N> symmetric_eigen(const Mat<N> &x) {
    Mat<N> a = x;
    Mat<N> v = identity<N>();
    for (int sweep = 0; sweep < 12; ++sweep) {
        float off = 0.f;
        for (int p = 0; p < N; ++p)
            for (int q = p + 1; q < N; ++q) off += a.c[q][p] * a.c[q][p];
        if (off == 0.f) break;
        for (int p = 0; p < N; ++p)
            for (int q = p + 1; q < N; ++q) {
                const float apq = a.c[q][p];
                if (apq == 0.f) continue;
                const float theta = (a.c[q][q] - a.c[p][p]) / (2.f * apq);
                const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
                const float cs = 1.f / sqrtf(t * t + 1.f), sn = t * cs;
                // A <- J^T A J on rows/columns p, q (A stays symmetric: update both triangles)
                for (int k = 0; k < N; ++k) {
                    const float akp = a.c[p][k], akq = a.c[q][k];
                    a.c[p][k] = cs * akp - sn * akq;
                    a.c[q][k] = sn * akp + cs * akq;
                }
                for (int k = 0; k < N; ++k) {
                    const float apk = a.c[k][p], aqk = a.c[k][q];
                    a.c[k][p] = cs * apk - sn * aqk;
                    a.c[k][q] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < N; ++k) {
                    const float vkp = v.c[p][k], vkq = v.c[q][k];
                    v.c[p][k] = cs * vkp - sn * vkq;
                    v.c[q][k] = sn * vkp + cs * vkq;
                }
            }
    }
    SymmetricEigen<N> r;
    r.eigenvectors = v;
    for (int i = 0; i < N; ++i) r.eigenvalues.v[i] = a.c[i][i];
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// SVD (svd2.wgsl, svd3.wgsl): M = U diag(S) Vt, S descending and non-negative. One-sided Jacobi (Hestenes) on the columns.
// ---------------------------------------------------------------------------------------------------------------
template <int N>
struct Svd {
    Mat<N> u;
    Vec<N> s;
    Mat<N> vt;
};
template <int N>
WGG_FN Svd<N> svd(const Mat<N> &x) {
    Mat<N> a = x;              // columns become U * S
    Mat<N> v = identity<N>();  // accumulates V
    for (int sweep = 0; sweep < 16; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < N; ++p)
            for (int q = p + 1; q < N; ++q) {
                float app = 0.f, aqq = 0.f, apq = 0.f;
                for (int k = 0; k < N; ++k) {
                    app += a.c[p][k] * a.c[p][k];
                    aqq += a.c[q][k] * a.c[q][k];
                    apq += a.c[p][k] * a.c[q][k];
                }
                if (fabsf(apq) <= 1e-7f * sqrtf(app * aqq) || apq == 0.f) continue; // columns already orthogonal to f32 precision
                rotated = true;
                const float theta = (aqq - app) / (2.f * apq);
                const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
                const float cs = 1.f / sqrtf(t * t + 1.f), sn = t * cs;
                for (int k = 0; k < N; ++k) {
                    const float ap = a.c[p][k], aq = a.c[q][k];
                    a.c[p][k] = cs * ap - sn * aq;
                    a.c[q][k] = sn * ap + cs * aq;
                    const float vp = v.c[p][k], vq = v.c[q][k];
                    v.c[p][k] = cs * vp - sn * vq;
                    v.c[q][k] = sn * vp + cs * vq;
                }
            }
        if (!rotated) break;
    }
    Svd<N> r;
    float s[N];
    for (int j = 0; j < N; ++j) {
        float n2 = 0.f;
        for (int k = 0; k < N; ++k) n2 += a.c[j][k] * a.c[j][k];
        s[j] = sqrtf(n2);
    }
    int order[N];
    for (int j = 0; j < N; ++j) order[j] = j;
    for (int i = 0; i < N; ++i) // selection sort, descending
        for (int j = i + 1; j < N; ++j)
            if (s[order[j]] > s[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    for (int j = 0; j < N; ++j) {
        const int src = order[j];
        r.s.v[j] = s[src];
        const float invs = s[src] > 0.f ? 1.f / s[src] : 0.f;
        for (int k = 0; k < N; ++k) {
            r.u.c[j][k] = a.c[src][k] * invs;
            r.vt.c[k][j] = v.c[src][k]; // Vt(j, k) = V(k, j)
        }
    }
    // rank-deficient input: complete the zero columns of U to an orthonormal basis (Gram-Schmidt against the canonical basis)
    for (int j = 0; j < N; ++j) {
        if (r.s.v[j] > 0.f) continue;
        for (int e = 0; e < N; ++e) {
            float w[N];
            for (int k = 0; k < N; ++k) w[k] = k == e ? 1.f : 0.f;
            for (int jj = 0; jj < N; ++jj) {
                if (jj == j) continue;
                float d = 0.f;
                for (int k = 0; k < N; ++k) d += r.u.c[jj][k] * w[k];
                for (int k = 0; k < N; ++k) w[k] -= d * r.u.c[jj][k];
            }
            float n2 = 0.f;
            for (int k = 0; k < N; ++k) n2 += w[k] * w[k];
            if (n2 > 1e-6f) {
                const float in = 1.f / sqrtf(n2);
                for (int k = 0; k < N; ++k) r.u.c[j][k] = w[k] * in;
                break;
            }
        }
    }
    return r;
}
template <int N>
WGG_FN Mat<N> recompose(const Svd<N> &d) { // svd2.wgsl:43-46, svd3.wgsl:309-312
    Mat<N> us = d.u;
    for (int j = 0; j < N; ++j)
        for (int k = 0; k < N; ++k) us.c[j][k] *= d.s.v[j];
    return mul(us, d.vt);
}

// ---------------------------------------------------------------------------------------------------------------
// Rot2 (rot2.wgsl): 2-D rotation stored as (cos, sin)
// ---------------------------------------------------------------------------------------------------------------
struct Rot2 {
    float cos, sin;
};
namespace rot2 {
WGG_FN Rot2 identity() { return Rot2{ 1.f, 0.f }; }
WGG_FN Rot2 fromAngle(float angle) { return Rot2{ cosf(angle), sinf(angle) }; }
WGG_FN float angle(Rot2 r) { return atan2f(r.sin, r.cos); }
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
} // namespace rot2

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
    const float hs = sinf(angle / 2.f), hc = cosf(angle / 2.f);
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
