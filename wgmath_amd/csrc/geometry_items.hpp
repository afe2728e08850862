// Per-item bodies of wg_geometry_apply, shared by the HIP kernels (geometry.hip) and the host build the CPU tests use
// (tests/cpp/geometry_host.cpp): one function per item kind, `__host__ __device__` like the header they exercise.
#pragma once
#include "../../include/wgebra_geometry.hpp"

namespace wgg_items {
using namespace wgebra::geometry;
enum { OP_INV = 0, OP_CHOLESKY = 1, OP_LU = 2, OP_QR = 3, OP_SYM_EIGEN = 4, OP_SVD = 5, OP_ROT2 = 6, OP_QUAT = 7, OP_SIM2 = 8, OP_SIM3 = 9,
       // the transform functions one by one on RAW coordinates (a quaternion that is not unit, a (cos, sin) pair that is no rotation): what the
       // fixtures executed from the reference's WGSL text hold (tests/golden/wgsl_exec_geometry.npz)
       OP_QUAT_RAW = 10, OP_ROT2_RAW = 11, OP_SIM2_RAW = 12, OP_SIM3_RAW = 13, OP_FROM = 14,
       // utils/trig.wgsl + utils/min_max.wgsl; the Rot2 functions the eigen-solvers use; eig2::eigenvalues; svd2/svd3::recompose (a matrix op: dim = 2, 3)
       OP_UTILS = 15, OP_ROT2_EXT = 16, OP_EIGVALS2 = 17, OP_SVD_RECOMPOSE = 18, OP_LAST = 18 };
WGG_FN bool is_mat_op(int op) { return op <= OP_SVD || op == OP_SVD_RECOMPOSE; }

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
WGG_FN void svd_item(int op, const float *in, float *o) { // the reference has svd2 and svd3 only
    if (op == OP_SVD) {
        const Svd<N> r = svd<N>(load_mat<N>(in));
        store_mat<N>(o, r.u);
        for (int k = 0; k < N; ++k) o[N * N + k] = r.s.v[k];
        store_mat<N>(o + N * N + N, r.vt);
    } else { // OP_SVD_RECOMPOSE: in = u, s, vt in the layout OP_SVD writes
        Svd<N> d;
        d.u = load_mat<N>(in);
        for (int k = 0; k < N; ++k) d.s.v[k] = in[N * N + k];
        d.vt = load_mat<N>(in + N * N + N);
        store_mat<N>(o, recompose<N>(d));
    }
}
template <>
WGG_FN void svd_item<4>(int, const float *, float *) {}

template <int N>
WGG_FN void mat_item(int op, const float *in, float *o) {
    if (op == OP_SVD || op == OP_SVD_RECOMPOSE) { svd_item<N>(op, in, o); return; }
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

WGG_FN void raw_item(int op, const float *p, float *o) {
    if (op == OP_QUAT_RAW) { // in: a (x, y, z, w), b, v;  out: mul(a, b), mulVec(a, v), invMulVec(a, v), toMatrix(a), renormalizeFast(a), inv(a)
        const Quat a{ p[0], p[1], p[2], p[3] }, b{ p[4], p[5], p[6], p[7] };
        const Vec<3> v{ { p[8], p[9], p[10] } };
        const Quat ab = quat::mul(a, b), rn = quat::renormalizeFast(a), ai = quat::inv(a);
        const Vec<3> r1 = quat::mulVec(a, v), r2 = quat::invMulVec(a, v);
        o[0] = ab.x; o[1] = ab.y; o[2] = ab.z; o[3] = ab.w;
        for (int k = 0; k < 3; ++k) { o[4 + k] = r1.v[k]; o[7 + k] = r2.v[k]; }
        store_mat<3>(o + 10, quat::toMatrix(a));
        o[19] = rn.x; o[20] = rn.y; o[21] = rn.z; o[22] = rn.w;
        o[23] = ai.x; o[24] = ai.y; o[25] = ai.z; o[26] = ai.w;
    } else if (op == OP_ROT2_RAW) { // in: a (cos, sin), b, v;  out: mul, mulVec, invMulVec, toMatrix, inv
        const Rot2 a{ p[0], p[1] }, b{ p[2], p[3] };
        const Vec<2> v{ { p[4], p[5] } };
        const Rot2 ab = rot2::mul(a, b), ai = rot2::inv(a);
        const Vec<2> r1 = rot2::mulVec(a, v), r2 = rot2::invMulVec(a, v);
        o[0] = ab.cos; o[1] = ab.sin; o[2] = r1.v[0]; o[3] = r1.v[1]; o[4] = r2.v[0]; o[5] = r2.v[1];
        store_mat<2>(o + 6, rot2::toMatrix(a));
        o[10] = ai.cos; o[11] = ai.sin;
    } else if (op == OP_SIM2_RAW) { // in: a (cos, sin, tx, ty, scale), b, pt;  out: mul, inv (5 each), mulPt, invMulPt, mulVec, invMulVec
        const Sim2 a{ Rot2{ p[0], p[1] }, Vec<2>{ { p[2], p[3] } }, p[4] }, b{ Rot2{ p[5], p[6] }, Vec<2>{ { p[7], p[8] } }, p[9] };
        const Vec<2> pt{ { p[10], p[11] } };
        const Sim2 ab = sim2::mul(a, b), ai = sim2::inv(a);
        const Sim2 *two[2] = { &ab, &ai };
        for (int k = 0; k < 2; ++k) {
            o[5 * k] = two[k]->rotation.cos; o[5 * k + 1] = two[k]->rotation.sin;
            o[5 * k + 2] = two[k]->translation.v[0]; o[5 * k + 3] = two[k]->translation.v[1]; o[5 * k + 4] = two[k]->scale;
        }
        const Vec<2> q1 = sim2::mulPt(a, pt), q2 = sim2::invMulPt(a, pt), q3 = sim2::mulVec(a, pt), q4 = sim2::invMulVec(a, pt);
        o[10] = q1.v[0]; o[11] = q1.v[1]; o[12] = q2.v[0]; o[13] = q2.v[1]; o[14] = q3.v[0]; o[15] = q3.v[1]; o[16] = q4.v[0]; o[17] = q4.v[1];
    } else if (op == OP_SIM3_RAW) { // in: a (q[4], t[3], scale), b, pt;  out: mul, inv (8 each), mulPt, invMulPt, mulVec, invMulVec
        const Sim3 a{ Quat{ p[0], p[1], p[2], p[3] }, Vec<3>{ { p[4], p[5], p[6] } }, p[7] };
        const Sim3 b{ Quat{ p[8], p[9], p[10], p[11] }, Vec<3>{ { p[12], p[13], p[14] } }, p[15] };
        const Vec<3> pt{ { p[16], p[17], p[18] } };
        const Sim3 ab = sim3::mul(a, b), ai = sim3::inv(a);
        const Sim3 *two[2] = { &ab, &ai };
        for (int k = 0; k < 2; ++k) {
            o[8 * k] = two[k]->rotation.x; o[8 * k + 1] = two[k]->rotation.y; o[8 * k + 2] = two[k]->rotation.z; o[8 * k + 3] = two[k]->rotation.w;
            for (int c = 0; c < 3; ++c) o[8 * k + 4 + c] = two[k]->translation.v[c];
            o[8 * k + 7] = two[k]->scale;
        }
        const Vec<3> q1 = sim3::mulPt(a, pt), q2 = sim3::invMulPt(a, pt), q3 = sim3::mulVec(a, pt), q4 = sim3::invMulVec(a, pt);
        for (int c = 0; c < 3; ++c) { o[16 + c] = q1.v[c]; o[19 + c] = q2.v[c]; o[22 + c] = q3.v[c]; o[25 + c] = q4.v[c]; }
    } else if (op == OP_FROM) { // in: scaled axis (3), angle;  out: quat::fromScaledAxis (4), rot2::fromAngle (2)
        const Quat q = quat::fromScaledAxis(Vec<3>{ { p[0], p[1], p[2] } });
        const Rot2 r = rot2::fromAngle(p[3]);
        o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; o[4] = r.cos; o[5] = r.sin;
    } else if (op == OP_UTILS) { // in: y, x, t, m[16];  out: stable_atan2(y, x), stable_tanh(t), then max / amax / max of the leading 2-, 3-, 4-sized prefix
        o[0] = trig::stable_atan2(p[0], p[1]);
        o[1] = trig::stable_tanh(p[2]);
        const float *m = p + 3;
        o[2] = min_max::max2(Vec<2>{ { m[0], m[1] } });
        o[3] = min_max::amax2x2(load_mat<2>(m));
        o[4] = min_max::max2x2(load_mat<2>(m));
        o[5] = min_max::max3(Vec<3>{ { m[0], m[1], m[2] } });
        o[6] = min_max::amax3x3(load_mat<3>(m));
        o[7] = min_max::max3x3(load_mat<3>(m));
        o[8] = min_max::max4(Vec<4>{ { m[0], m[1], m[2], m[3] } });
        o[9] = min_max::amax4x4(load_mat<4>(m));
        o[10] = min_max::max4x4(load_mat<4>(m));
    } else if (op == OP_ROT2_EXT) { // in: rot (cos, sin), v, m3[9], m4[16], i3, i4;  out: angle(rot), cancel_y(v), is_valid(cancel_y(v)), rotate_rows3, rotate_rows4
        const Rot2 r{ p[0], p[1] };
        const Rot2 cy = rot2::cancel_y(Vec<2>{ { p[2], p[3] } });
        o[0] = rot2::angle(r);
        o[1] = cy.cos; o[2] = cy.sin;
        o[3] = rot2::is_valid(cy) ? 1.f : 0.f;
        Mat3 m3 = load_mat<3>(p + 4);
        Mat4 m4 = load_mat<4>(p + 13);
        // the row indices come straight from caller data: only 0 .. N - 2 name two rows of the matrix (negative, NaN or larger values -- whose cast would be
        // undefined, and whose use an out-of-bounds write on the thread-local matrix -- leave it as it is; the reference's WGSL is memory-safe there)
        if (p[29] >= 0.f && p[29] < 2.f) rot2::rotate_rows3(r, m3, (uint32_t)p[29]);
        if (p[30] >= 0.f && p[30] < 3.f) rot2::rotate_rows4(r, m4, (uint32_t)p[30]);
        store_mat<3>(o + 4, m3);
        store_mat<4>(o + 13, m4);
    } else if (op == OP_EIGVALS2) { // in: a 2 x 2 symmetric matrix;  out: eig2::eigenvalues
        const Vec<2> e = eig2::eigenvalues(load_mat<2>(p));
        o[0] = e.v[0]; o[1] = e.v[1];
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
    case OP_QUAT_RAW: return 27;
    case OP_ROT2_RAW: return 12;
    case OP_SIM2_RAW: return 18;
    case OP_SIM3_RAW: return 28;
    case OP_FROM: return 6;
    case OP_UTILS: return 11;
    case OP_ROT2_EXT: return 29;
    case OP_EIGVALS2: return 2;
    case OP_SVD_RECOMPOSE: return n * n;
    }
    return 0;
}
WGG_FN unsigned in_floats(int op, unsigned n) {
    switch (op) {
    case OP_ROT2: return 4;
    case OP_QUAT: return 9;
    case OP_SIM2: return 10;
    case OP_SIM3: return 17;
    case OP_QUAT_RAW: return 11;
    case OP_ROT2_RAW: return 6;
    case OP_SIM2_RAW: return 12;
    case OP_SIM3_RAW: return 19;
    case OP_FROM: return 4;
    case OP_UTILS: return 19;
    case OP_ROT2_EXT: return 31;
    case OP_EIGVALS2: return 4;
    case OP_SVD_RECOMPOSE: return 2 * n * n + n;
    default: return n * n;
    }
}
} // namespace wgg_items
