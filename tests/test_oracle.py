"""CPU tests of the oracle itself (no GPU): the two restatements agree bit-for-bit, reproduce the committed golden
vectors, and clear the reference's own test bars on the reference's own test shapes."""
import os

import numpy as np
import pytest

import _util as U
from oracle import wgsl_oracle as wo


@pytest.fixture(scope="module")
def C():
    return wo.CLib()


# ---- the reference's own tests, run on the oracle (this is the pin the oracle has; see DESIGN.md "Oracle") ---------
def test_reference_gpu_gemm_procedure(C):
    """gemm.rs:149-200: 256x256 U[0,1), four variants, assert_relative_eq!(gpu, nalgebra, epsilon = 1e-3)."""
    rng = np.random.default_rng(1)
    n = 256
    m1, m2 = rng.random((n, n), dtype=np.float32), rng.random((n, n), dtype=np.float32)
    f1, f2 = m1.reshape(-1, order="F"), m2.reshape(-1, order="F")
    for variant in (wo.GEMM, wo.GEMM_TR, wo.GEMM_FAST, wo.GEMM_TR_FAST):
        out = np.zeros(n * n, np.float32)
        C.gemm(variant, out, wo.Shape(n, n), f1, wo.Shape(n, n), f2, wo.Shape(n, n))
        a = m1.T if variant in (wo.GEMM_TR, wo.GEMM_TR_FAST) else m1
        cpu = a @ m2  # what nalgebra computes, in f32
        assert U.relative_eq(out.reshape(n, n, order="F"), cpu, epsilon=1e-3)
        truth, sabs = wo.gemm_f64(a, m2)
        U.assert_close_f64(out.reshape(n, n, order="F"), truth, n, sabs, f"oracle gemm v{variant}")


def test_reference_gpu_gemv_procedure(C):
    """gemv.rs:158-195: 1024x1024, out pre-filled with random data, abs <= 1e-3."""
    rng = np.random.default_rng(2)
    n = 1024
    m, v = rng.random((n, n), dtype=np.float32), rng.random(n, dtype=np.float32)
    for variant in (wo.GEMV, wo.GEMV_TR, wo.GEMV_FAST, wo.GEMV_TR_FAST):
        out = rng.random(n, dtype=np.float32)
        C.gemv(variant, out, wo.Shape(n), m.reshape(-1, order="F"), wo.Shape(n, n), v, wo.Shape(n))
        a = m.T if variant in (wo.GEMV_TR, wo.GEMV_TR_FAST) else m
        assert U.relative_eq(out, a @ v, epsilon=1e-3)


def test_reference_gpu_reduce_procedure(C):
    """reduce.rs:143-177: 345 U[0,1) numbers, five ops, abs <= 1e-3 vs nalgebra min/max/sum/norm_squared/product."""
    rng = np.random.default_rng(3)
    x = rng.random(345, dtype=np.float32)
    ref = {wo.MIN: x.min(), wo.MAX: x.max(), wo.SUM: x.sum(), wo.SQNORM: (x * x).sum(), wo.PROD: x.prod()}
    for op, expect in ref.items():
        assert U.relative_eq(C.reduce(op, x, wo.Shape(345)), expect, epsilon=1e-3)


def test_reference_gpu_op_assign_known_answer(C):
    """op_assign.rs:110-155: the one deterministic vector pair the reference holds; its expectation is IEEE f32 + - * /."""
    g = U.golden("op_assign_ref_1757")
    v0 = (np.arange(1757, dtype=np.float32) + np.float32(0.1)).astype(np.float32)
    v1 = (np.arange(1757, dtype=np.float32) * np.float32(10.0) + np.float32(0.1)).astype(np.float32)
    U.assert_bits_equal(g["v0"], v0)
    U.assert_bits_equal(g["v1"], v1)
    with np.errstate(all="ignore"):
        cpu = {wo.ADD: v0 + v1, wo.SUB: v0 - v1, wo.MUL: v0 * v1, wo.DIV: v0 / v1, wo.COPY: v1}
    for op, expect in cpu.items():
        a = v0.copy()
        C.op_assign(op, a, wo.Shape(1757), v1, wo.Shape(1757))
        U.assert_bits_equal(a, expect.astype(np.float32), f"oracle op_assign {op}")
        U.assert_bits_equal(a, g[f"expected_{op}"], f"golden op_assign {op}")
        assert U.relative_eq(a, expect, epsilon=1e-7)  # the reference's literal bar


# ---- golden vectors ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["gemm_u01_64x256x32x2", "gemm_pm1_64x256x32x2"])
def test_golden_gemm(C, name):
    g = U.golden(name)
    M, K, N, mats = (int(x) for x in g["dims"])
    for variant in range(4):
        tr = variant in (wo.GEMM_TR, wo.GEMM_TR_FAST)
        key = "tr" if tr else "nn"
        s1 = wo.Shape(K, M, mats) if tr else wo.Shape(M, K, mats)
        for impl in (C.gemm, wo.gemm):
            out = np.zeros(M * N * mats, np.float32)
            impl(variant, out, wo.Shape(M, N, mats), g[f"m1_{key}"], s1, g[f"m2_{key}"], wo.Shape(K, N, mats))
            U.assert_bits_equal(out, g[f"out_v{variant}"], f"{name} v{variant} {impl.__module__}")
        U.assert_close_f64(g[f"out_v{variant}"], g[f"truth_{key}"], K, g[f"sabs_{key}"], f"{name} v{variant} vs f64")


@pytest.mark.parametrize("name", ["gemv_u01_128x256x3x2", "gemv_pm1_128x256x3x2"])
def test_golden_gemv(C, name):
    g = U.golden(name)
    R, Cc, nrhs, mats = (int(x) for x in g["dims"])
    for variant in range(4):
        tr = variant in (wo.GEMV_TR, wo.GEMV_TR_FAST)
        key = "tr" if tr else "nn"
        vlen, olen = (R, Cc) if tr else (Cc, R)
        for impl in (C.gemv, wo.gemv):
            out = np.full(olen * nrhs * mats, 7.0, np.float32)
            impl(variant, out, wo.Shape(olen, nrhs, mats), g["m"], wo.Shape(R, Cc, mats), g[f"v_{key}"], wo.Shape(vlen, nrhs, mats))
            U.assert_bits_equal(out, g[f"out_v{variant}"], f"{name} v{variant}")
        U.assert_close_f64(g[f"out_v{variant}"], g[f"truth_{key}"], vlen, g[f"sabs_{key}"], f"{name} v{variant} vs f64")


def test_golden_reduce(C):
    g = U.golden("reduce")
    for n in (0, 1, 127, 128, 129, 345, 65536, "prod4096"):
        x = g[f"x_{n}"]
        for op in range(5):
            got_np = wo.reduce(op, x, wo.Shape(x.size))
            U.assert_bits_equal(np.float32(got_np), g[f"expected_{n}"][op], f"numpy reduce n={n} op={op}")
            if x.size:
                U.assert_bits_equal(np.float32(C.reduce(op, x, wo.Shape(x.size))), g[f"expected_{n}"][op], f"C reduce n={n} op={op}")
    assert list(g["expected_0"]) == [np.float32(3.4e38), np.float32(-3.4e38), 0.0, 1.0, 0.0]  # reduce.wgsl:40-46 init values
    for op in range(5):
        U.assert_bits_equal(C.reduce_batched(op, g["xb"], wo.Shape(1000, 96)), g[f"expected_batched_{op}"])


# ---- the two restatements agree on more shapes, incl. strided views and batches -----------------------------------------
@pytest.mark.parametrize("M,K,N,mats", [(4, 4, 4, 1), (8, 256, 12, 2), (36, 512, 28, 1), (64, 20, 32, 3)])
def test_c_vs_numpy_gemm(C, M, K, N, mats):
    rng = np.random.default_rng(M + K + N)
    for variant in range(4):
        if variant in (wo.GEMM_FAST, wo.GEMM_TR_FAST) and K % 256:
            continue  # the fast kernels have no tail guard (gemm.wgsl:40,162)
        tr = variant in (wo.GEMM_TR, wo.GEMM_TR_FAST)
        # embed the operands in larger parents: stride > rows, offset != 0
        pr = 8
        s1 = wo.Shape(K, M, mats, K + pr, (K + pr) * M + 16, 4) if tr else wo.Shape(M, K, mats, M + pr, (M + pr) * K + 16, 4)
        s2 = wo.Shape(K, N, mats, K + 4, (K + 4) * N + 8, 8)
        so = wo.Shape(M, N, mats, M + 12, (M + 12) * N + 4, 12)
        ext = lambda s: s.resolved().stride_mat * mats + s.offset + 64
        m1 = rng.random(ext(s1), dtype=np.float32) - np.float32(0.5)
        m2 = rng.random(ext(s2), dtype=np.float32) - np.float32(0.5)
        o1 = rng.random(ext(so), dtype=np.float32)
        o2 = o1.copy()
        C.gemm(variant, o1, so, m1, s1, m2, s2)
        wo.gemm(variant, o2, so, m1, s1, m2, s2)
        U.assert_bits_equal(o1, o2, f"gemm v{variant} {M}x{K}x{N}x{mats} strided")


def test_oracle_error_paths(C):
    z = np.zeros(64, np.float32)
    with pytest.raises(wo.OracleError) as e:
        C.gemm(wo.GEMM, z, wo.Shape(8, 8), z, wo.Shape(8, 4), z, wo.Shape(8, 8))
    assert e.value.code == wo.ERR_DIM
    with pytest.raises(wo.OracleError) as e:
        C.gemv(wo.GEMV_FAST, z, wo.Shape(6), z, wo.Shape(6, 8), z, wo.Shape(8))
    assert e.value.code == wo.ERR_ASSERT
    with pytest.raises(wo.OracleError) as e:  # the kernel would read past the end of m1
        C.gemm(wo.GEMM, z, wo.Shape(8, 8), z[:32], wo.Shape(8, 8), z, wo.Shape(8, 8))
    assert e.value.code == wo.ERR_OOB
    with pytest.raises(wo.OracleError) as e:
        C.op_assign(wo.ADD, z, wo.Shape(8), z, wo.Shape(12))
    assert e.value.code == wo.ERR_DIM
    # GemvTrFast with rows % 128 != 0 falls back to GemvTr (gemv.rs:99-104): identical bits
    rng = np.random.default_rng(5)
    m, v = rng.random(8 * 12, dtype=np.float32), rng.random(8, dtype=np.float32)
    o1, o2 = np.zeros(12, np.float32), np.zeros(12, np.float32)
    C.gemv(wo.GEMV_TR_FAST, o1, wo.Shape(12), m, wo.Shape(8, 12), v, wo.Shape(8))
    C.gemv(wo.GEMV_TR, o2, wo.Shape(12), m, wo.Shape(8, 12), v, wo.Shape(8))
    U.assert_bits_equal(o1, o2)
    # zero-sized bindings are silently skipped (kernel.rs:111-123)
    e0 = np.zeros(0, np.float32)
    C.op_assign(wo.ADD, e0, wo.Shape(0), e0, wo.Shape(0))


def test_oracle_sampled_grid_matches_full(C):
    """The cpu_baseline leg times a slice of the grid (wg_begin/wg_end): the slice must compute exactly those rows."""
    rng = np.random.default_rng(9)
    M, K, N = 512, 64, 16
    a, b = rng.random(M * K, dtype=np.float32), rng.random(K * N, dtype=np.float32)
    full, part = np.zeros(M * N, np.float32), np.full(M * N, -1.0, np.float32)
    C.gemm(wo.GEMM, full, wo.Shape(M, N), a, wo.Shape(M, K), b, wo.Shape(K, N))
    C.gemm(wo.GEMM, part, wo.Shape(M, N), a, wo.Shape(M, K), b, wo.Shape(K, N), 1, 2)  # invocations 64..127 -> rows 256..511
    f, p = full.reshape(M, N, order="F"), part.reshape(M, N, order="F")
    assert np.array_equal(p[256:512], f[256:512]) and (p[:256] == -1.0).all()


# ------------------------------------------------------------------------------------------------------------
# Pinning against the reference's OWN shader text: tests/golden/wgsl_exec_*.npz hold the outputs of the reference's .wgsl files
# executed by oracle/wgsl_exec.py (generator: tests/golden/make_wgsl_golden.py, run where /root/reference is mounted).
# Both restatements must reproduce them bit for bit.
# ------------------------------------------------------------------------------------------------------------
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _shape_of(row):
    return wo.Shape(*[int(x) for x in row])


def test_restatement_matches_executed_wgsl_gemm(oracle_c):
    g = np.load(os.path.join(GOLD, "wgsl_exec_gemm.npz"))
    keys = sorted({k[: -len("_out")] for k in g.files if k.endswith("_out")})
    assert len(keys) >= 13
    variants = {"gemm": wo.GEMM, "gemm_fast": wo.GEMM_FAST, "gemm_tr": wo.GEMM_TR, "gemm_tr_fast": wo.GEMM_TR_FAST}
    for k in keys:
        entry = k.split("_", 1)[1]
        variant = variants[entry]
        so, s1, s2 = (_shape_of(r) for r in g[k + "_shapes"])
        m1, m2, exp = g[k + "_m1"], g[k + "_m2"], g[k + "_out"]
        for name, impl in (("numpy", wo.gemm), ("C", oracle_c.gemm)):
            out = np.full(exp.size, np.nan, np.float32)
            impl(variant, out, so, m1, s1, m2, s2)
            assert out.tobytes() == exp.tobytes(), f"{name} restatement of {entry} ({k}) differs from the executed WGSL"


def test_restatement_matches_executed_wgsl_gemv(oracle_c):
    g = np.load(os.path.join(GOLD, "wgsl_exec_gemv.npz"))
    keys = sorted({k[: -len("_out")] for k in g.files if k.endswith("_out")})
    assert len(keys) >= 10
    variants = {"gemv": wo.GEMV, "gemv_fast": wo.GEMV_FAST, "gemv_tr": wo.GEMV_TR, "gemv_tr_fast": wo.GEMV_TR_FAST}
    for k in keys:
        entry = k.split("_", 1)[1]
        so, sm, sv = (_shape_of(r) for r in g[k + "_shapes"])
        m, v, exp = g[k + "_m"], g[k + "_v"], g[k + "_out"]
        for name, impl in (("numpy", wo.gemv), ("C", oracle_c.gemv)):
            out = np.full(exp.size, np.nan, np.float32)
            impl(variants[entry], out, so, m, sm, v, sv)
            assert out.tobytes() == exp.tobytes(), f"{name} restatement of {entry} ({k}) differs from the executed WGSL"


def test_restatement_matches_executed_wgsl_reduce_and_op_assign(oracle_c):
    g = np.load(os.path.join(GOLD, "wgsl_exec_reduce.npz"))
    keys = sorted({k[: -len("_res")] for k in g.files if k.endswith("_res")})
    assert len(keys) == 35
    for k in keys:
        op, n, off = int(k.split("_")[0][2:]), int(k.split("_")[1][1:]), int(k.split("_")[2][1:])
        x, exp = g[k + "_x"], g[k + "_res"]
        sh = wo.Shape(n, 1, 1, n, n, off)
        assert np.float32(wo.reduce(op, x, sh)).tobytes() == exp[0].tobytes(), f"numpy reduce op {op} n {n}"
        assert np.float32(oracle_c.reduce(op, x, sh)).tobytes() == exp[0].tobytes(), f"C reduce op {op} n {n}"
    g = np.load(os.path.join(GOLD, "wgsl_exec_op_assign.npz"))
    keys = sorted({k[: -len("_a0")] for k in g.files if k.endswith("_a0")})
    assert len(keys) == 15
    for k in keys:
        op, n, oa, ob = int(k.split("_")[0][2:]), int(k.split("_")[1][1:]), int(k.split("_")[2]), int(k.split("_")[3])
        a0, b, exp = g[k + "_a0"], g[k + "_b"], g[k + "_a"]
        for name, impl in (("numpy", wo.op_assign), ("C", oracle_c.op_assign)):
            a = a0.copy()
            impl(op, a, wo.Shape(n, 1, 1, n, n, oa), b, wo.Shape(n, 1, 1, n, n, ob))
            assert a.tobytes() == exp.tobytes(), f"{name} op_assign op {op} ({k}) differs from the executed WGSL"


def test_index_math_matches_executed_wgsl():
    """shape.wgsl's `it` / `iv` / `with_vec4_elts`, column-major and ROW_MAJOR, executed from the reference's text, against the
    formulas this repo uses (column-major: the oracle's Shape; row-major: index = t*stride_mat + offset + i*stride + j, the
    semantics of wg_gemm_rm / wg_gemv_rm)."""
    g = np.load(os.path.join(GOLD, "wgsl_exec_shape.npz"))
    for tag in ("cm", "rm"):
        for r in g[tag]:
            nrows, ncols, nmats, stride, stride_mat, offset, i, j, t, it, iv, v_nr, v_nc, v_nm, v_st, v_sm, v_off = (int(x) for x in r)
            assert iv == offset + i
            if tag == "cm":
                assert it == t * stride_mat + offset + i + j * stride
                assert (v_nr, v_nc) == ((nrows + 3) // 4, ncols)
            else:
                assert it == t * stride_mat + offset + i * stride + j
                assert (v_nr, v_nc) == (nrows, (ncols + 3) // 4)
            assert (v_nm, v_st, v_sm, v_off) == (nmats, (stride + 3) // 4, (stride_mat + 3) // 4, offset // 4)
