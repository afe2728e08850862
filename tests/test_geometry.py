"""Geometry header (include/wgebra_geometry.hpp): the reference's own test procedure for its geometry shaders
(crates/wgebra/src/geometry/*.rs: LEN = 345 random matrices per size through a one-item-per-invocation kernel, results compared with
nalgebra at relative eps 1e-3 / 1e-4, 1-2 % of badly conditioned samples allowed to miss) with NumPy/f64 in nalgebra's place.
The same checks run on the host build of the header (CPU suite) and on the HIP kernels through wg_geometry_apply (-m gpu)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LEN = 345
OPS = dict(INV=0, CHOLESKY=1, LU=2, QR=3, SYM_EIGEN=4, SVD=5, ROT2=6, QUAT=7, SIM2=8, SIM3=9, QUAT_RAW=10, ROT2_RAW=11, SIM2_RAW=12, SIM3_RAW=13, FROM=14,
           UTILS=15, ROT2_EXT=16, EIGVALS2=17, SVD_RECOMPOSE=18)


def out_floats(op, n):
    return {0: n * n, 1: n * n, 2: n * n + 2 * n + 1, 3: 2 * n * n, 4: n * n + n, 5: 2 * n * n + n, 6: 11, 7: 19, 8: 14, 9: 25, 10: 27, 11: 12, 12: 18, 13: 28, 14: 6,
            15: 11, 16: 29, 17: 2, 18: n * n}[op]


def in_floats(op, n):
    return {6: 4, 7: 9, 8: 10, 9: 17, 10: 11, 11: 6, 12: 12, 13: 19, 14: 4, 15: 19, 16: 31, 17: 4, 18: 2 * n * n + n}.get(op, n * n)


@pytest.fixture(scope="module")
def host_apply():
    build = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(build, exist_ok=True)
    so = os.path.join(build, "libgeom_host.so")
    src = os.path.join(ROOT, "tests", "cpp", "geometry_host.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src], check=True)
    lib = ctypes.CDLL(so)
    lib.geom_apply_host.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]

    def apply(op, dim, inp):
        inp = np.ascontiguousarray(inp, np.float32)
        count = inp.shape[0]
        out = np.zeros((count, out_floats(op, dim)), np.float32)
        assert lib.geom_apply_host(op, dim, inp.ctypes.data, out.ctypes.data, count) == 0
        return out
    return apply


def gpu_apply_factory(gpu):
    import wgmath_amd as wg
    S = wg.BufferUsages

    def apply(op, dim, inp):
        inp = np.ascontiguousarray(inp, np.float32)
        count = inp.shape[0]
        tin = wg.TensorBuilder.vector(inp.size, S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(gpu.device(), inp.reshape(-1))
        tout = wg.TensorBuilder.vector(count * out_floats(op, dim), S.STORAGE | S.COPY_SRC).build(gpu.device(), np.float32)
        assert wg.geometry.in_floats(op, dim) == in_floats(op, dim) and in_floats(op, dim) * count == inp.size
        assert wg.geometry.out_floats(op, dim) == out_floats(op, dim)
        wg.geometry.apply(gpu, wg.GeomOp(op), dim, tin, tout, count)
        return tout.read(gpu.device()).reshape(count, -1)
    return apply


def cm(a, n):  # (count, n*n) column-major items -> (count, n, n) matrices
    return a.reshape(-1, n, n).transpose(0, 2, 1)


def rel_close(a, b, eps):
    """approx::relative_eq on matrices: |a - b| <= eps * max(|a|, |b|) elementwise (or absolutely tiny)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= np.maximum(eps * np.maximum(np.abs(a), np.abs(b)), 1e-6), axis=tuple(range(1, a.ndim)))


def allow(ok, pct, what):
    fails = int((~ok).sum())
    assert fails <= max(1, len(ok) * pct // 100), f"{what}: {fails}/{len(ok)} items miss the reference's tolerance"


def run_checks(apply):
    rng = np.random.default_rng(2024)
    for n in (2, 3, 4):
        M = rng.random((LEN, n, n)).astype(np.float32)                      # nalgebra `new_random`: U[0, 1)
        items = M.transpose(0, 2, 1).reshape(LEN, n * n)                    # column-major items
        M64 = M.astype(np.float64)
        # inverse (inv.rs): against the f64 inverse, eps 1e-3; ill-conditioned samples allowed to miss
        got = cm(apply(OPS["INV"], n, items), n)
        allow(rel_close(got, np.linalg.inv(M64), 1e-3) | (np.linalg.cond(M64) > 1e3), 2, f"inv{n}")
        # Cholesky (cholesky.rs:99-131): SPD = M^T M; lower triangle vs the f64 factor, upper triangle untouched
        spd = np.einsum("bki,bkj->bij", M, M).astype(np.float32)
        got = cm(apply(OPS["CHOLESKY"], n, spd.transpose(0, 2, 1).reshape(LEN, n * n)), n)
        ref = np.linalg.cholesky(spd.astype(np.float64))
        low = np.tril(np.ones((n, n), bool))
        allow(rel_close(np.where(low, got, 0), ref, 1e-3) | (np.linalg.cond(spd.astype(np.float64)) > 1e4), 2, f"cholesky{n}")
        assert np.array_equal(np.where(low, 0, got), np.where(low, 0, spd)), "cholesky: upper triangle must keep the input"
        # LU (lu.rs:150-181): lu_internal of partial-pivoting LU; P A = L U with the recorded swaps
        o = apply(OPS["LU"], n, items)
        lu, ia, ib, ln = cm(o[:, :n * n], n).astype(np.float64), o[:, n * n:n * n + n].astype(int), o[:, n * n + n:n * n + 2 * n].astype(int), o[:, -1].astype(int)
        L = np.tril(lu, -1) + np.eye(n)
        U = np.triu(lu)
        PA = M64.copy()
        for b in range(LEN):
            for k in range(ln[b]):
                PA[b, [ia[b, k], ib[b, k]]] = PA[b, [ib[b, k], ia[b, k]]]
        allow(rel_close(L @ U, PA, 1e-3), 1, f"lu{n}")
        assert (np.abs(np.tril(lu, -1)) <= 1 + 1e-6).all(), "partial pivoting: |L| <= 1"
        # QR (qr3.rs:100-118): Q orthonormal, R upper triangular with non-negative diagonal, Q R = M; unique => equals nalgebra's
        o = apply(OPS["QR"], n, items)
        Q, R = cm(o[:, :n * n], n).astype(np.float64), cm(o[:, n * n:], n).astype(np.float64)
        q_ref, r_ref = np.linalg.qr(M64)
        sg = np.sign(np.diagonal(r_ref, axis1=1, axis2=2)); sg[sg == 0] = 1
        q_ref, r_ref = q_ref * sg[:, None, :], r_ref * sg[:, :, None]
        allow(rel_close(Q, q_ref, 1e-3) & rel_close(R, r_ref, 1e-3), 2, f"qr{n}")
        assert (np.tril(R, -1) == 0).all() and (np.diagonal(R, axis1=1, axis2=2) >= 0).all()
        # symmetric eigen (eig3.rs:108-127): reconstruction V diag(l) V^T == M, eps 1e-4, 2 % may miss
        sym = ((M + M.transpose(0, 2, 1)) * 0.5).astype(np.float32)
        o = apply(OPS["SYM_EIGEN"], n, sym.transpose(0, 2, 1).reshape(LEN, n * n))
        V, lam = cm(o[:, :n * n], n).astype(np.float64), o[:, n * n:].astype(np.float64)
        rec = np.einsum("bik,bk,bjk->bij", V, lam, V)
        allow(rel_close(rec, sym, 1e-4), 2, f"symmetric_eigen{n}")
        allow(rel_close(np.einsum("bki,bkj->bij", V, V), np.broadcast_to(np.eye(n), (LEN, n, n)), 1e-4), 2, f"eigenvectors{n} orthonormal")
        assert np.allclose(np.sort(lam, axis=1), np.linalg.eigvalsh(sym.astype(np.float64)), atol=1e-4)
        # SVD (svd2.rs:96-106, svd3.rs:100-110): U diag(S) Vt == M at eps 1e-4 for EVERY item (the reference's SVD tests allow no misses); 2 and 3 only,
        # like the reference. |S| descending; svd3's last singular value may come out negative (the sign stays in S: svd3.wgsl:303), svd2's never does.
        if n < 4:
            o = apply(OPS["SVD"], n, items)
            U_, S_, Vt = cm(o[:, :n * n], n).astype(np.float64), o[:, n * n:n * n + n].astype(np.float64), cm(o[:, n * n + n:], n).astype(np.float64)
            assert rel_close(np.einsum("bik,bk,bkj->bij", U_, S_, Vt), M64, 1e-4).all(), f"svd{n}"
            assert (S_[:, :n - 1] >= 0).all() and (np.diff(np.abs(S_), axis=1) <= 1e-6).all() and (n == 3 or (S_ >= 0).all())
            assert np.allclose(np.abs(S_), np.linalg.svd(M64, compute_uv=False), atol=1e-4)
            rec = cm(apply(OPS["SVD_RECOMPOSE"], n, o), n)
            assert rel_close(rec, M64, 1e-4).all(), f"svd{n}::recompose"
    # Rot2 / Quat / Sim2 / Sim3 against NumPy rotation matrices
    ang = (rng.random((LEN, 2)) * 2 - 1).astype(np.float32) * 3
    vec = (rng.random((LEN, 2)) * 2 - 1).astype(np.float32)
    o = apply(OPS["ROT2"], 0, np.concatenate([ang, vec], 1)).astype(np.float64)
    rot = lambda t: np.stack([np.stack([np.cos(t), -np.sin(t)], -1), np.stack([np.sin(t), np.cos(t)], -1)], -2)
    Ra, Rb = rot(ang[:, 0].astype(np.float64)), rot(ang[:, 1].astype(np.float64))
    Rab = Ra @ Rb
    assert np.allclose(o[:, 0], Rab[:, 0, 0], atol=1e-5) and np.allclose(o[:, 1], Rab[:, 1, 0], atol=1e-5)
    assert np.allclose(o[:, 2:4], np.einsum("bij,bj->bi", Ra, vec), atol=1e-5)
    assert np.allclose(o[:, 4:6], np.einsum("bji,bj->bi", Ra, vec), atol=1e-5)
    assert np.allclose(cm(o[:, 6:10], 2), Ra, atol=1e-6)
    assert np.allclose(np.cos(o[:, 10]), Rab[:, 0, 0], atol=1e-5) and np.allclose(np.sin(o[:, 10]), Rab[:, 1, 0], atol=1e-5)

    def rodrigues(aa):
        th = np.linalg.norm(aa, axis=1)[:, None, None]
        k = aa / np.maximum(np.linalg.norm(aa, axis=1, keepdims=True), 1e-30)
        Kx = np.zeros((len(aa), 3, 3))
        Kx[:, 0, 1], Kx[:, 0, 2], Kx[:, 1, 0], Kx[:, 1, 2], Kx[:, 2, 0], Kx[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
        return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)

    def quat_to_mat(q):
        x, y, z, w = q.T
        return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                         np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                         np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)
    aa = ((rng.random((LEN, 6)) * 2 - 1) * 2).astype(np.float32)
    v3 = (rng.random((LEN, 3)) * 2 - 1).astype(np.float32)
    o = apply(OPS["QUAT"], 0, np.concatenate([aa, v3], 1)).astype(np.float64)
    Qa, Qb = rodrigues(aa[:, :3].astype(np.float64)), rodrigues(aa[:, 3:].astype(np.float64))
    assert np.allclose(quat_to_mat(o[:, 0:4]), Qa @ Qb, atol=2e-5)
    assert np.allclose(o[:, 4:7], np.einsum("bij,bj->bi", Qa, v3), atol=2e-5)
    assert np.allclose(o[:, 7:10], np.einsum("bji,bj->bi", Qa, v3), atol=2e-5)
    assert np.allclose(cm(o[:, 10:19], 3), Qa, atol=2e-5)
    # Sim2: x -> s R x + t
    p2 = np.concatenate([ang[:, :1], vec, (0.5 + rng.random((LEN, 1))).astype(np.float32), ang[:, 1:], vec[:, ::-1], (0.5 + rng.random((LEN, 1))).astype(np.float32),
                         (rng.random((LEN, 2)) * 2 - 1).astype(np.float32)], 1).astype(np.float32)
    o = apply(OPS["SIM2"], 0, p2).astype(np.float64)
    p = p2.astype(np.float64)
    sa, ta, sb, tb, pt = p[:, 3:4], p[:, 1:3], p[:, 7:8], p[:, 5:7], p[:, 8:10]
    Ra, Rb = rot(p[:, 0]), rot(p[:, 4])
    f_a = lambda x: sa * np.einsum("bij,bj->bi", Ra, x) + ta
    f_b = lambda x: sb * np.einsum("bij,bj->bi", Rb, x) + tb
    apply_sim2 = lambda row, x: row[:, 3:4] * np.einsum("bij,bj->bi", rot(row[:, 0]), x) + row[:, 1:3]
    assert np.allclose(apply_sim2(o[:, 0:4], pt), f_a(f_b(pt)), atol=5e-5)           # mul composes
    assert np.allclose(apply_sim2(o[:, 4:8], f_a(pt)), pt, atol=5e-5)                # inv undoes
    assert np.allclose(o[:, 8:10], f_a(pt), atol=5e-5)                               # mulPt
    assert np.allclose(f_a(o[:, 10:12]), pt, atol=5e-5)                              # invMulPt
    assert np.allclose(o[:, 12:14], sa * np.einsum("bij,bj->bi", Ra, pt), atol=5e-5)  # mulVec (no translation)
    # Sim3
    t3 = (rng.random((LEN, 6)) * 2 - 1).astype(np.float32)
    sc = (0.5 + rng.random((LEN, 2))).astype(np.float32)
    p3 = np.concatenate([aa[:, :3], t3[:, :3], sc[:, :1], aa[:, 3:], t3[:, 3:], sc[:, 1:], v3], 1).astype(np.float32)
    o = apply(OPS["SIM3"], 0, p3).astype(np.float64)
    p = p3.astype(np.float64)
    Qa, Qb = rodrigues(p[:, 0:3]), rodrigues(p[:, 7:10])
    g_a = lambda x: p[:, 6:7] * np.einsum("bij,bj->bi", Qa, x) + p[:, 3:6]
    g_b = lambda x: p[:, 13:14] * np.einsum("bij,bj->bi", Qb, x) + p[:, 10:13]
    apply_sim3 = lambda row, x: row[:, 7:8] * np.einsum("bij,bj->bi", quat_to_mat(row[:, 0:4]), x) + row[:, 4:7]
    pt = p[:, 14:17]
    assert np.allclose(apply_sim3(o[:, 0:8], pt), g_a(g_b(pt)), atol=1e-4)
    assert np.allclose(apply_sim3(o[:, 8:16], g_a(pt)), pt, atol=1e-4)
    assert np.allclose(o[:, 16:19], g_a(pt), atol=1e-4)
    assert np.allclose(g_a(o[:, 19:22]), pt, atol=1e-4)
    assert np.allclose(o[:, 22:25], p[:, 6:7] * np.einsum("bij,bj->bi", Qa, pt), atol=1e-4)


def test_geometry_header_host(host_apply):
    run_checks(host_apply)


@pytest.mark.gpu
def test_geometry_gpu_matches_reference_procedure(gpu):
    run_checks(gpu_apply_factory(gpu))


def same_bits(got, exp):
    """Bit-for-bit, NaN where NaN (payloads aside). Returns (ok, description of the first difference)."""
    got, exp = np.asarray(got, np.float32), np.asarray(exp, np.float32)
    nan_g, nan_e = np.isnan(got), np.isnan(exp)
    bad = (nan_g != nan_e) | (~nan_g & ~nan_e & (got.view(np.uint32) != exp.view(np.uint32)))
    if not bad.any():
        return True, ""
    at = tuple(np.argwhere(bad)[0])
    return False, f"{int(bad.sum())} values differ, first at item/field {at}: {got[at]!r} vs {exp[at]!r}"


@pytest.mark.gpu
def test_geometry_gpu_vs_host(gpu, host_apply):
    """The HIP build and the host build of the same header return the same bits on every item kind: contraction is off in both, division and sqrt are
    correctly rounded in both, sin / cos / atan / exp are the float64 functions rounded once in both."""
    gapply = gpu_apply_factory(gpu)
    rng = np.random.default_rng(7)
    for op, dims in ((0, (2, 3, 4)), (1, (2, 3, 4)), (2, (2, 3, 4)), (3, (2, 3, 4)), (4, (2, 3, 4)), (5, (2, 3)), (6, (0,)), (7, (0,)), (8, (0,)), (9, (0,)),
                     (10, (0,)), (11, (0,)), (12, (0,)), (13, (0,)), (14, (0,)), (15, (0,)), (17, (0,)), (18, (2, 3))):
        for n in dims:
            nin = in_floats(op, n)
            x = (rng.random((512, nin)) * 2 - 1).astype(np.float32) + (1.5 if op in (8, 9, 12, 13) else 0)
            if op in (1, 4) and n:  # SPD / symmetric inputs
                m = x.reshape(-1, n, n)
                m = np.einsum("bki,bkj->bij", m, m) + np.eye(n, dtype=np.float32) if op == 1 else (m + m.transpose(0, 2, 1)) / 2
                x = m.reshape(-1, n * n).astype(np.float32)
            ok, why = same_bits(gapply(op, n, x), host_apply(op, n, x))
            assert ok, f"op {op} dim {n}: GPU and host builds differ: {why}"


# --------------------------------------------------------------------------------------------------------
# Pinned to the reference's WGSL TEXT: tests/golden/wgsl_exec_geometry.npz and wgsl_exec_decomp.npz hold seeded inputs and what the reference's
# crates/wgebra/src/geometry/*.wgsl and utils/{trig,min_max}.wgsl return when executed by oracle/wgsl_exec.py (left to right, every product and sum
# rounded, dot / cross / length / matrix products as their defining formulas, `fma` one rounding, sin / cos / atan / exp correctly rounded; generators:
# tests/golden/make_wgsl_geometry_golden.py, make_wgsl_decomp_golden.py). The header follows the same statements with contraction off: 0 ulp on EVERY
# function, no allowance -- non-unit quaternions, (cos, sin) pairs that are no rotations, and the inputs on which the reference itself returns NaN
# included (svd3 of a matrix whose sorted first column starts with two exact zeros: rsqrt1(0) overflows to inf * -0 (svd3.wgsl:69-80); eig4 when the
# Wilkinson shift divides by d + sign(d) * .. with d == 0, sign(0) = 0 (eig3.wgsl:205) -- the fixtures hold them and the header returns NaN there too).
# --------------------------------------------------------------------------------------------------------
def max_ulp(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia, ib = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia), np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return int(np.abs(ia - ib).max())


def run_wgsl_pinned(apply):
    g = np.load(os.path.join(ROOT, "tests", "golden", "wgsl_exec_geometry.npz"))
    for n in (2, 3, 4):
        for op in ("INV", "CHOLESKY", "LU"):
            key = f"{op.lower()}{n}"
            got = apply(OPS[op], n, g[key + "_in"])
            assert got.tobytes() == g[key + "_out"].tobytes(), f"{key}: {max_ulp(got, g[key + '_out'])} ulp from the executed WGSL (must be 0)"
    for op, key in (("QUAT_RAW", "quat_raw"), ("ROT2_RAW", "rot2_raw"), ("SIM2_RAW", "sim2_raw"), ("SIM3_RAW", "sim3_raw"), ("FROM", "from")):
        ok, why = same_bits(apply(OPS[op], 0, g[key + "_in"]), g[key + "_out"])
        assert ok, f"{key} vs the executed WGSL: {why}"
    assert np.array_equal(apply(OPS["FROM"], 0, g["from_in"])[0, :4], np.array([0, 0, 0, 1], np.float32))  # the zero axis is the identity (quat.wgsl:20-22)


def run_wgsl_pinned_decompositions(apply):
    """QR, symmetric eigen, SVD (+ recompose), eig2::eigenvalues, trig, min_max and the Rot2 helpers against the executed reference text: 0 ulp."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "wgsl_exec_decomp.npz"))
    seen = 0
    for n in (2, 3, 4):
        for op, key in (("QR", f"qr{n}"), ("SYM_EIGEN", f"sym_eigen{n}"), ("SVD", f"svd{n}"), ("SVD_RECOMPOSE", f"svd_recompose{n}")):
            if key + "_in" not in g.files:
                assert n == 4 and op.startswith("SVD")
                continue
            ok, why = same_bits(apply(OPS[op], n, g[key + "_in"]), g[key + "_out"])
            assert ok, f"{key} vs the executed WGSL: {why}"
            seen += 1
    for op, key in (("EIGVALS2", "eigvals2"), ("UTILS", "utils"), ("ROT2_EXT", "rot2_ext")):
        ok, why = same_bits(apply(OPS[op], 0, g[key + "_in"]), g[key + "_out"])
        assert ok, f"{key} vs the executed WGSL: {why}"
    assert seen == 10
    # the conventions the reference's text fixes, read off the fixture itself (so a regenerated fixture cannot silently drop them)
    e2_in, e2 = g["sym_eigen2_in"], g["sym_eigen2_out"]
    off = e2_in[:, 1] != 0
    assert (e2[off, 4] >= e2[off, 5]).all(), "eig2: ((a + b + sigma) / 2, (a + b - sigma) / 2) in that order (eig2.wgsl:28-31)"
    assert (e2[off][:, [1, 3]] > 0).all(), "eig2: eigenvectors normalised with last component + (eig2.wgsl:32-35)"
    assert np.array_equal(e2[~off, :4], np.tile(np.array([1, 0, 0, 1], np.float32), ((~off).sum(), 1))), "eig2: c == 0 returns the identity basis"
    ut_in, ut = g["utils_in"], g["utils_out"]
    axis = (ut_in[:, 1] == 0) | ((ut_in[:, 1] < 0) & (ut_in[:, 0] == 0))
    assert axis.sum() >= 6 and (ut[axis, 0] == 0).all(), "stable_atan2 is 0 for x == 0 and for x < 0, y == 0 (trig.wgsl:25-38)"
    for n in (2, 3, 4):
        r = g[f"qr{n}_out"][:, n * n:].reshape(-1, n, n)     # columns: r[:, c, r]
        assert (np.diagonal(r, axis1=1, axis2=2) >= 0).all() and (np.triu(r.transpose(0, 2, 1), 0) == r.transpose(0, 2, 1)).all()
    assert np.isnan(g["svd3_out"]).any() and np.isnan(g["sym_eigen4_out"]).any()  # the reference's own NaN cases are in the fixture


def test_closed_form_functions_match_the_executed_wgsl_host(host_apply):
    run_wgsl_pinned(host_apply)


def test_decompositions_and_utils_match_the_executed_wgsl_host(host_apply):
    run_wgsl_pinned_decompositions(host_apply)


@pytest.mark.gpu
def test_closed_form_functions_match_the_executed_wgsl_gpu(gpu):
    run_wgsl_pinned(gpu_apply_factory(gpu))


@pytest.mark.gpu
def test_decompositions_and_utils_match_the_executed_wgsl_gpu(gpu):
    run_wgsl_pinned_decompositions(gpu_apply_factory(gpu))
