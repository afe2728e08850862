// Per-item bodies of wg_geometry_apply, shared by the HIP kernels (geometry.hip) and the host build the CPU tests use
// (tests/cpp/geometry_host.cpp): one function per item kind, `__host__ __device__` like the header they exercise.
#pragma once
#include "../../include/wgebra_geometry.hpp"

namespace wgg_items {
using namespace wgebra::geometry;
enum { OP_INV = 0, OP_CHOLESKY = 1, OP_LU = 2, OP_QR = 3, OP_SYM_EIGEN = 4, OP_SVD = 5, OP_ROT2 = 6, OP_QUAT = 7, OP_SIM2 = 8, OP_SIM3 = 9 };

template <int N>
WGG_FN Mat<N> load_mat(const float *p) {
    Mat<N> m;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) m.c[j][i] = p[j * N + i];
    return m;
}
template <int N>
WGG_FN void store_mat(float *p, const Mat<N> &m) {
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) p[j * N + i] = m.c[j][i];
}

template <int N>
WGG_FN void mat_item(int op, const float *in, float *o) {
    const Mat<N> m = load_mat<N>(in);
    switch (op) {
    case OP_INV: store_mat<N>(o, inv<N>(m)); break;
    case OP_CHOLESKY: store_mat<N>(o, cholesky<N>(m)); break;
    case OP_LU: {
        const LU<N> r = lu<N>(m);
        store_mat<N>(o, r.lu);
        for (int k = 0; k < N; ++k) { o[N * N + k] = (float)r.p.ia[k]; o[N * N + N + k] = (float)r.p.ib[k]; }
        o[N * N + 2 * N] = (float)r.p.len;
    } break;
    case OP_QR: {
        const QR<N> r = qr<N>(m);
        store_mat<N>(o, r.q);
        store_mat<N>(o + N * N, r.r);
    } break;
    case OP_SYM_EIGEN: {
        const SymmetricEigen<N> r = symmetric_eigen<N>(m);
        store_mat<N>(o, r.eigenvectors);
        for (int k = 0; k < N; ++k) o[N * N + k] = r.eigenvalues.v[k];
    } break;
    case OP_SVD: {
        const Svd<N> r = svd<N>(m);
        store_mat<N>(o, r.u);
        for (int k = 0; k < N; ++k) o[N * N + k] = r.s.v[k];
        store_mat<N>(o + N * N + N, r.vt);
    } break;
    }
}

WGG_FN void transform_item(int op, const float *p, float *o) {
    if (op == OP_ROT2) {
        const Rot2 a = rot2::fromAngle(p[0]), b = rot2::fromAngle(p[1]);
        const Vec<2> v{ { p[2], p[3] } };
        const Rot2 ab = rot2::mul(a, b);
        const Vec<2> r1 = rot2::mulVec(a, v), r2 = rot2::invMulVec(a, v);
        o[0] = ab.cos; o[1] = ab.sin; o[2] = r1.v[0]; o[3] = r1.v[1]; o[4] = r2.v[0]; o[5] = r2.v[1];
        store_mat<2>(o + 6, rot2::toMatrix(a));
        o[10] = rot2::angle(ab);
    } else if (op == OP_QUAT) {
        const Quat a = quat::fromScaledAxis(Vec<3>{ { p[0], p[1], p[2] } }), b = quat::fromScaledAxis(Vec<3>{ { p[3], p[4], p[5] } });
        const Vec<3> v{ { p[6], p[7], p[8] } };
        const Quat ab = quat::mul(a, b);
        const Vec<3> r1 = quat::mulVec(a, v), r2 = quat::invMulVec(a, v);
        o[0] = ab.x; o[1] = ab.y; o[2] = ab.z; o[3] = ab.w;
        for (int k = 0; k < 3; ++k) { o[4 + k] = r1.v[k]; o[7 + k] = r2.v[k]; }
        store_mat<3>(o + 10, quat::toMatrix(a));
    } else if (op == OP_SIM2) {
        const Sim2 a{ rot2::fromAngle(p[0]), Vec<2>{ { p[1], p[2] } }, p[3] }, b{ rot2::fromAngle(p[4]), Vec<2>{ { p[5], p[6] } }, p[7] };
        const Vec<2> pt{ { p[8], p[9] } };
        const Sim2 ab = sim2::mul(a, b), ai = sim2::inv(a);
        o[0] = rot2::angle(ab.rotation); o[1] = ab.translation.v[0]; o[2] = ab.translation.v[1]; o[3] = ab.scale;
        o[4] = rot2::angle(ai.rotation); o[5] = ai.translation.v[0]; o[6] = ai.translation.v[1]; o[7] = ai.scale;
        const Vec<2> q1 = sim2::mulPt(a, pt), q2 = sim2::invMulPt(a, pt), q3 = sim2::mulVec(a, pt);
        o[8] = q1.v[0]; o[9] = q1.v[1]; o[10] = q2.v[0]; o[11] = q2.v[1]; o[12] = q3.v[0]; o[13] = q3.v[1];
    } else if (op == OP_SIM3) {
        const Sim3 a{ quat::fromScaledAxis(Vec<3>{ { p[0], p[1], p[2] } }), Vec<3>{ { p[3], p[4], p[5] } }, p[6] };
        const Sim3 b{ quat::fromScaledAxis(Vec<3>{ { p[7], p[8], p[9] } }), Vec<3>{ { p[10], p[11], p[12] } }, p[13] };
        const Vec<3> pt{ { p[14], p[15], p[16] } };
        const Sim3 ab = sim3::mul(a, b), ai = sim3::inv(a);
        o[0] = ab.rotation.x; o[1] = ab.rotation.y; o[2] = ab.rotation.z; o[3] = ab.rotation.w;
        for (int k = 0; k < 3; ++k) o[4 + k] = ab.translation.v[k];
        o[7] = ab.scale;
        o[8] = ai.rotation.x; o[9] = ai.rotation.y; o[10] = ai.rotation.z; o[11] = ai.rotation.w;
        for (int k = 0; k < 3; ++k) o[12 + k] = ai.translation.v[k];
        o[15] = ai.scale;
        const Vec<3> q1 = sim3::mulPt(a, pt), q2 = sim3::invMulPt(a, pt), q3 = sim3::mulVec(a, pt);
        for (int k = 0; k < 3; ++k) { o[16 + k] = q1.v[k]; o[19 + k] = q2.v[k]; o[22 + k] = q3.v[k]; }
    }
}

WGG_FN unsigned out_floats(int op, unsigned n) {
    switch (op) {
    case OP_INV: case OP_CHOLESKY: return n * n;
    case OP_LU: return n * n + 2 * n + 1;
    case OP_QR: return 2 * n * n;
    case OP_SYM_EIGEN: return n * n + n;
    case OP_SVD: return 2 * n * n + n;
    case OP_ROT2: return 11;
    case OP_QUAT: return 19;
    case OP_SIM2: return 14;
    case OP_SIM3: return 25;
    }
    return 0;
}
WGG_FN unsigned in_floats(int op, unsigned n) {
    switch (op) {
    case OP_ROT2: return 4;
    case OP_QUAT: return 9;
    case OP_SIM2: return 10;
    case OP_SIM3: return 17;
    default: return n * n;
    }
}
} // namespace wgg_items
