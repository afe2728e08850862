"""GPU parity tests proper: HIP kernels (through the C ABI) vs the oracle and the committed golden vectors.
Sizes here are ones the oracle finishes in seconds; BASELINE-size runs live in test_gpu_fullsize.py."""
import numpy as np
import pytest

import _util as U

pytestmark = pytest.mark.gpu


def _wg():
    import wgmath_amd as wg
    return wg


def _wo():
    from oracle import wgsl_oracle as wo
    return wo


S_STORAGE = 128 | 4 | 8  # STORAGE | COPY_SRC | COPY_DST


def upload(gpu, shape, flat, dtype=np.float32):
    wg = _wg()
    return wg.TensorBuilder.tensor(shape, S_STORAGE).build_init(gpu.device(), np.asarray(flat, dtype), dtype)


def run_pass(gpu, fn):
    enc = gpu.device().create_command_encoder()
    with enc.compute_pass("test", None) as p:
        fn(p)
    gpu.queue().submit([enc.finish()])


# --------------------------------------------------------------------------------------------------------
# golden vectors
# --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["gemm_u01_64x256x32x2", "gemm_pm1_64x256x32x2"])
def test_gemm_golden(gpu, name):
    wg = _wg()
    g = U.golden(name)
    M, K, N, mats = (int(x) for x in g["dims"])
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    for variant in wg.GemmVariant:
        tr = variant in (wg.GemmVariant.GemmTr, wg.GemmVariant.GemmTrFast)
        key = "tr" if tr else "nn"
        m1 = upload(gpu, (K, M, mats) if tr else (M, K, mats), g[f"m1_{key}"])
        m2 = upload(gpu, (K, N, mats), g[f"m2_{key}"])
        out = upload(gpu, (M, N, mats), np.full(M * N * mats, np.nan, np.float32))
        run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, m1, m2, variant))
        got = out.read(gpu.device())
        U.assert_close_f64(got, g[f"truth_{key}"], K, g[f"sabs_{key}"], f"{name} {variant!r} vs f64")
        U.assert_close_oracle(got, g[f"out_v{int(variant)}"], K, g[f"sabs_{key}"], f"{name} {variant!r} vs golden")


@pytest.mark.parametrize("name", ["gemv_u01_128x256x3x2", "gemv_pm1_128x256x3x2"])
def test_gemv_golden(gpu, name):
    wg = _wg()
    g = U.golden(name)
    R, C, nrhs, mats = (int(x) for x in g["dims"])
    gemv, shapes = wg.Gemv.from_device(gpu.device()), wg.ViewShapeBuffers()
    m = upload(gpu, (R, C, mats), g["m"])
    for variant in wg.GemvVariant:
        tr = variant in (wg.GemvVariant.GemvTr, wg.GemvVariant.GemvTrFast)
        key = "tr" if tr else "nn"
        vlen, olen = (R, C) if tr else (C, R)
        v = upload(gpu, (vlen, nrhs, mats), g[f"v_{key}"])
        out = upload(gpu, (olen, nrhs, mats), np.full(olen * nrhs * mats, np.nan, np.float32))
        run_pass(gpu, lambda p: gemv.dispatch_generic(gpu.device(), shapes, p, out, m, v, variant))
        got = out.read(gpu.device())
        U.assert_close_f64(got, g[f"truth_{key}"], vlen, g[f"sabs_{key}"], f"{name} {variant!r} vs f64")
        U.assert_close_oracle(got, g[f"out_v{int(variant)}"], vlen, g[f"sabs_{key}"], f"{name} {variant!r} vs golden")


def test_reduce_golden_bit_exact(gpu):
    wg = _wg()
    g = U.golden("reduce")
    shapes = wg.ViewShapeBuffers()
    for n in (0, 1, 127, 128, 129, 345, 65536, "prod4096"):
        x = g[f"x_{n}"]
        vec = upload(gpu, (max(x.size, 1),), x if x.size else np.zeros(1, np.float32))
        for op in wg.ReduceOp:
            res = upload(gpu, (), np.array([np.nan], np.float32))
            red = wg.Reduce.new(gpu.device(), op)
            view = vec.as_view().rows(0, x.size) if x.size else vec.rows(0, 0)
            run_pass(gpu, lambda p: red.dispatch(gpu.device(), shapes, p, view, res))
            got = res.read(gpu.device())
            U.assert_bits_equal(got, g[f"expected_{n}"][int(op):int(op) + 1], f"reduce n={n} {op!r}")


def test_reduce_batched_and_unaligned(gpu, oracle_c):
    wg, wo = _wg(), _wo()
    g = U.golden("reduce")
    shapes = wg.ViewShapeBuffers()
    xb = g["xb"]
    mat = upload(gpu, (1000, 96), xb)
    for op in wg.ReduceOp:
        red = wg.Reduce.new(gpu.device(), op)
        res = upload(gpu, (96,), np.full(96, np.nan, np.float32))
        run_pass(gpu, lambda p: red.dispatch_batched(gpu.device(), shapes, p, mat, res))
        U.assert_bits_equal(res.read(gpu.device()), g[f"expected_batched_{int(op)}"], f"batched reduce {op!r}")
        # unaligned vector base (offset 3, length 777): the kernel at an element-aligned base must give the same bits as the oracle
        res1 = upload(gpu, (), np.array([np.nan], np.float32))
        view = wg.GpuTensorView(wg.ViewShape((777, 1, 1), 1, 1, 3), mat, 1)
        run_pass(gpu, lambda p: red.dispatch(gpu.device(), shapes, p, view, res1))
        exp = oracle_c.reduce(int(op), xb, wo.Shape(777, 1, 1, 1, 1, 3))
        U.assert_bits_equal(res1.read(gpu.device()), np.array([exp], np.float32), f"unaligned reduce {op!r}")
        # odd stride (not a multiple of 4) -> every column unaligned, batched
        v2 = wg.GpuTensorView(wg.ViewShape((333, 50, 1), 1001, 1, 2), mat, 2)
        res2 = upload(gpu, (50,), np.full(50, np.nan, np.float32))
        run_pass(gpu, lambda p: red.dispatch_batched(gpu.device(), shapes, p, v2, res2))
        exp2 = oracle_c.reduce_batched(int(op), xb, wo.Shape(333, 50, 1, 1001, 1, 2))
        U.assert_bits_equal(res2.read(gpu.device()), exp2, f"odd-stride batched reduce {op!r}")


# --------------------------------------------------------------------------------------------------------
# shape sweeps vs the oracle on the same seeded inputs
# --------------------------------------------------------------------------------------------------------
GEMM_SHAPES = [
    # M,   K,   N, mats
    (4, 4, 4, 1), (8, 4, 12, 1), (36, 20, 28, 1), (256, 256, 128, 1), (260, 264, 132, 1), (512, 48, 260, 2),
    (128, 1024, 64, 1), (1024, 16, 512, 1), (252, 1000, 124, 3),
    # few output tiles + long K -> split-K path (f32 slabs + ordered reduce), incl. ragged K and batches
    (256, 4096, 128, 1), (512, 8192, 256, 1), (260, 2064, 132, 2), (64, 16384, 64, 1),
    # tail split: 17 x 31 = 527 and 17 x 33 = 561 tiles on 512 resident workgroups (full and ragged tiles)
    (4352, 512, 3968, 1), (4104, 256, 4104, 1),
    # few columns (N <= 64, M >= 512, K >= 128): the streaming MFMA kernel of gemm_f32_skinny.hip (NN; GemmTr stays on the tiled kernel)
    (4096, 4096, 16, 1), (1024, 516, 64, 2), (520, 132, 4, 1), (2048, 1000, 36, 1), (516, 128, 32, 3), (8192, 260, 8, 1),
    # few rows (M <= 64, N >= 512, K >= 128): computed transposed (GemmTr with few columns + transposes of the small operands)
    (16, 4096, 4096, 1), (64, 260, 1000, 2), (8, 128, 516, 1), (36, 1024, 2048, 1), (4, 512, 8192, 3),
    # K % 16 != 0 on interior tiles (extra zero-filled k-tile after the pipelined loop), alone, under split-K and under the tail split
    (512, 36, 256, 1), (768, 1044, 384, 2), (256, 4100, 128, 1), (4104, 268, 4104, 1), (3328, 520, 3328, 1),
    # more than one wave of 2 x CUs tiles with a leftover (1280 = 2.5 waves; 17 x 83 ragged): the leftover tiles cut along K as their own launch
    (4096, 256, 10240, 1), (4100, 272, 10500, 1),
    # 65 .. 128 rows and more than 4096 columns: computed transposed on the tiled kernel (transposed copy of m1 for Gemm, transposed result)
    (128, 256, 4224, 1), (72, 132, 5000, 2), (100, 1028, 4352, 1),
]


@pytest.mark.parametrize("M,K,N,mats", [(64, 64, 64, 9), (32, 32, 32, 12), (96, 200, 48, 5), (256, 256, 256, 3), (36, 44, 20, 7), (132, 36, 68, 4), (40, 32, 16, 6)])
@pytest.mark.parametrize("tr", [False, True])
def test_gemm_small_batched_on_panels(gpu, oracle_c, M, K, N, mats, tr):
    """Batches of small f32 matrices run on the 64-column panels of the few-column kernel when the launch plan expects that to pay (hundreds of
    matrices); forced here (wg_ctx_set_tuning) so that a handful of matrices exercises the same kernel, and once more on the plan's own choice."""
    for force in (1, -1):
        old = gpu.set_tuning("f32_panels", force)
        try:
            test_gemm_shapes(gpu, oracle_c, M, K, N, mats, tr)
        finally:
            gpu.set_tuning("f32_panels", old)


@pytest.mark.parametrize("M,K,N,mats", GEMM_SHAPES)
@pytest.mark.parametrize("tr", [False, True])
def test_gemm_shapes(gpu, oracle_c, M, K, N, mats, tr):
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(M * 1000003 + K * 1009 + N * 7 + mats + int(tr))
    a = (rng.random(M * K * mats, dtype=np.float32) * 2 - 1).astype(np.float32)
    b = (rng.random(K * N * mats, dtype=np.float32) * 2 - 1).astype(np.float32)
    s1 = wo.Shape(K, M, mats) if tr else wo.Shape(M, K, mats)
    s2, so = wo.Shape(K, N, mats), wo.Shape(M, N, mats)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    orc = np.zeros(M * N * mats, np.float32)
    oracle_c.gemm(int(variant), orc, so, a, s1, b, s2)
    m1 = upload(gpu, (K, M, mats) if tr else (M, K, mats), a)
    m2 = upload(gpu, (K, N, mats), b)
    out = upload(gpu, (M, N, mats), np.full(M * N * mats, np.nan, np.float32))
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, m1, m2, variant))
    got = out.read(gpu.device())
    A, B = wo.view(a, s1), wo.view(b, s2)
    for t in range(mats):
        amk = A[:, :, t].T if tr else A[:, :, t]
        truth, sabs = wo.gemm_f64(amk, B[:, :, t])
        g_t = wo.view(got, so)[:, :, t]
        U.assert_close_f64(g_t, truth, K, sabs, f"gemm {M}x{K}x{N} mat {t} tr={tr} vs f64")
        U.assert_close_oracle(g_t, wo.view(orc, so)[:, :, t], K, sabs, "vs oracle")


def test_gemm_identity_asymmetric(gpu):
    """A = I with an asymmetric B catches a transposed/permuted C write (guide: 'always A=I-check with asymmetric B')."""
    wg = _wg()
    M = K = 384
    N = 136
    eye = np.eye(M, K, dtype=np.float32)
    B = (np.arange(K * N, dtype=np.float32).reshape(K, N) % 1021) + np.arange(N, dtype=np.float32)[None, :] * 0.5
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    for variant in (wg.GemmVariant.Gemm, wg.GemmVariant.GemmTr):
        m1, m2 = upload(gpu, (M, K), eye), upload(gpu, (K, N), B)
        out = upload(gpu, (M, N), np.zeros(M * N, np.float32))
        run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, m1, m2, variant))
        got = out.read(gpu.device()).reshape(M, N, order="F")
        assert np.array_equal(got, B.astype(np.float32)), variant


def test_gemm_strided_views(gpu, oracle_c):
    """Views built by GpuMatrix::columns / rows / GpuCubeView::matrix (tensor.rs:466-626): non-default stride/offset."""
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(77)
    PR, PC = 96, 80  # parents
    pa = (rng.random(PR * PC * 2, dtype=np.float32) - 0.5).astype(np.float32)
    pb = (rng.random(PR * PC * 2, dtype=np.float32) - 0.5).astype(np.float32)
    po0 = (rng.random(PR * PC * 2, dtype=np.float32)).astype(np.float32)
    ta, tb, to = upload(gpu, (PR, PC, 2), pa), upload(gpu, (PR, PC, 2), pb), upload(gpu, (PR, PC, 2), po0)
    # m1 = rows 8..72 (64) x cols 4..36 (32) of matrix 1 of ta; m2 = rows 0..32 x cols 8..28 (20) of matrix 1 of tb
    a_view = ta.as_view().matrix(1).columns(4, 32).rows(8, 64)
    b_view = tb.as_view().matrix(1).columns(8, 20).rows(0, 32)
    o_view = to.as_view().matrix(0).columns(12, 20).rows(16, 64)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    run_pass(gpu, lambda p: gemm.dispatch(gpu.device(), shapes, p, o_view, a_view, b_view))
    got = to.read(gpu.device())
    orc = po0.copy()
    sh = lambda v: wo.Shape(v.shape().size[0], v.shape().size[1], v.shape().size[2], v.shape().stride, v.shape().stride_mat, v.shape().offset)
    oracle_c.gemm(0, orc, sh(o_view), pa, sh(a_view), pb, sh(b_view))
    A, B = wo.view(pa, sh(a_view))[:, :, 0], wo.view(pb, sh(b_view))[:, :, 0]
    truth, sabs = wo.gemm_f64(A, B)
    U.assert_close_f64(wo.view(got, sh(o_view))[:, :, 0], truth, 32, sabs, "strided gemm vs f64")
    # everything outside the output view is untouched
    mask = np.ones(po0.size, bool)
    s = sh(o_view).resolved()
    idx = s.offset + np.arange(s.nrows)[:, None] + np.arange(s.ncols)[None, :] * s.stride
    mask[idx.ravel()] = False
    assert np.array_equal(got[mask], po0[mask])
    U.assert_close_oracle(wo.view(got, sh(o_view))[:, :, 0], wo.view(orc, sh(o_view))[:, :, 0], 32, sabs, "strided gemm vs oracle")


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("tr", [False, True])
def test_gemm_unaligned_views(gpu, dtype, tr):
    """Views that are not vec4-aligned -- what GpuMatrix::slice((1, 0), ..), rows(1, n), columns of a parent with an odd row count and
    lengths that are not multiples of 4 produce (tensor.rs:574-626) -- which the reference's vec4 kernels cannot address
    (shape.wgsl:64-66). They compute op(A) B like any other view here (lengths off a multiple of 4: on zero-padded copies): against f64 with the usual bound, nothing outside the output
    view touched, alpha / beta honoured."""
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(321 + int(tr))
    PR, PC = 101, 90  # parents with an odd leading dimension
    pa = (rng.random(PR * PC, dtype=np.float32) - 0.5).astype(dtype)
    pb = (rng.random(PR * PC, dtype=np.float32) - 0.5).astype(dtype)
    po0 = rng.random(PR * PC, dtype=np.float32).astype(dtype)
    ta, tb, to = upload(gpu, (PR, PC), pa, dtype), upload(gpu, (PR, PC), pb, dtype), upload(gpu, (PR, PC), po0, dtype)
    M, K, N = 37, 26, 19
    # rows(1, ..) then columns: offset 1 + 3 * 101, stride 101, odd lengths
    a_view = wg.GpuTensorView(wg.ViewShape(((K, M) if tr else (M, K)) + (1,), PR, PR * PC, 1 + 3 * PR), ta, 2)
    b_view = tb.slice((1, 0), (K, N))                      # the reference's slice: offset i + j * nrows = 1, stride = parent rows
    o_view = wg.GpuTensorView(wg.ViewShape((M, N, 1), PR, PR * PC, 2 + 5 * PR), to, 2)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, o_view, a_view, b_view, variant))
    got = to.read(gpu.device())
    sh = lambda v: wo.Shape(v.shape().size[0], v.shape().size[1], v.shape().size[2], v.shape().stride, v.shape().stride_mat, v.shape().offset)
    A, B = wo.view(pa, sh(a_view))[:, :, 0].astype(np.float64), wo.view(pb, sh(b_view))[:, :, 0].astype(np.float64)
    A = A.T if tr else A
    truth, sabs = A @ B, np.abs(A) @ np.abs(B)
    G = wo.view(got, sh(o_view))[:, :, 0].astype(np.float64)
    tol = U.f32_gate(K, sabs) + (2.0 ** -11 * np.abs(truth) + 2.0 ** -25 if dtype == np.float16 else 0.0)
    assert (np.abs(G - truth) <= tol).all(), f"unaligned gemm: worst err/tol {(np.abs(G - truth) / tol).max():.3g}"
    mask = np.ones(po0.size, bool)
    s_ = sh(o_view).resolved()
    mask[(s_.offset + np.arange(M)[:, None] + np.arange(N)[None, :] * s_.stride).ravel()] = False
    assert np.array_equal(got[mask], po0[mask]), "unaligned gemm wrote outside its output view"
    # alpha / beta through the same path (wg_gemm_ex): out = 0.5 * A B - 2 * out
    from wgmath_amd import _lib
    to2 = upload(gpu, (PR, PC), po0, dtype)
    o2 = wg.GpuTensorView(o_view.shape(), to2, 2)
    _lib.check(_lib.lib.wg_gemm_ex(gpu._ctx.handle, int(variant), wg.wgcore.wg_dtype(dtype), 0.5, -2.0, to2._h, o2.shape().to_c(), ta._h, a_view.shape().to_c(),
                                   tb._h, b_view.shape().to_c()))
    got2 = wo.view(to2.read(gpu.device()), sh(o_view))[:, :, 0].astype(np.float64)
    C0 = wo.view(po0, sh(o_view))[:, :, 0].astype(np.float64)
    want = 0.5 * truth - 2.0 * C0
    tol2 = 0.5 * tol + 4 * 2.0 ** (-11 if dtype == np.float16 else -24) * (np.abs(want) + 2 * np.abs(C0)) + 1e-6
    assert (np.abs(got2 - want) <= tol2).all()


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_copy_view_every_alignment_on_both_sides(gpu, dtype):
    """wg_copy_view (the staging pass of the unaligned Gemm / Gemv paths, transpose.hip copy2d_kernel): every combination of source and destination
    offset mod 16 bytes, odd leading dimensions (a different shift in every column), narrow and wide columns, a source smaller than the
    destination (zero fill), a batch -- the destination view holds exactly the source or 0, nothing outside it is touched."""
    wg = _wg()
    from wgmath_amd import _lib
    rng = np.random.default_rng(2025)
    per = 16 // np.dtype(dtype).itemsize
    cases = []
    for so in range(per):
        for do in range(per):
            cases.append((37, 5, 37, 5, 41, 43, so, do, 1))          # narrow columns, odd leading dimensions
    cases += [(1000, 3, 1003, 4, 1001, 1005 + d, s, d, 1) for s in (0, 1, 3) for d in (0, 2, per - 1)]  # wide columns, zero fill on both axes
    cases += [(300, 2, 300, 2, 304, 304, 0, 0, 3), (300, 2, 290, 2, 301, 299, 5, 3, 2), (64, 7, 64, 9, 64, 64, 0, 4 % per, 1), (5, 3, 8, 3, 5, 8, 1, 0, 1)]
    for (rs, cs, rd, cd, lds, ldd, so, do, mats) in cases:
        src = (rng.random(so + lds * max(cs, 1) * mats + 16, dtype=np.float32) - 0.5).astype(dtype)
        dst0 = (rng.random(do + ldd * cd * mats + 16, dtype=np.float32) + 1.0).astype(dtype)
        ts, td = upload(gpu, (src.size,), src, dtype), upload(gpu, (dst0.size,), dst0, dtype)
        sv = wg.ViewShape((rs, cs, mats), lds, lds * cs, so)
        dv = wg.ViewShape((rd, cd, mats), ldd, ldd * cd, do)
        _lib.check(_lib.lib.wg_copy_view(gpu._ctx.handle, wg.wgcore.wg_dtype(dtype), td._h, dv.to_c(), ts._h, sv.to_c()))
        got = td.read(gpu.device())
        want = dst0.copy()
        for z in range(mats):
            for j in range(cd):
                col = np.zeros(rd, dtype)
                if j < cs:
                    n = min(rs, rd)
                    col[:n] = src[so + z * lds * cs + j * lds: so + z * lds * cs + j * lds + n]
                want[do + z * ldd * cd + j * ldd: do + z * ldd * cd + j * ldd + rd] = col
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (rs, cs, rd, cd, lds, ldd, so, do, mats)


F16_ANY_ALIGN = [
    # M, N, K, matrices -- one shape per f16 kernel family / launch plan (gemm_f16.hip launcher): 128 x 128 tiles; one 256 x 128 tile per CU; the continuous walk;
    # per-tile launches with a K % 64 remainder; split-K slabs + reduce; full rounds + a K-cut tail; a batch; few columns
    (512, 512, 512, 1), (4096, 2048, 1024, 1), (4096, 4096, 512, 1), (2048, 2304, 328, 1), (256, 256, 8192, 1), (4352, 4096, 1024, 1), (1024, 768, 256, 3), (8192, 16, 1024, 1),
]


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("case", F16_ANY_ALIGN)
def test_gemm_f16_any_offset_and_leading_dimension_is_the_aligned_product_bit_for_bit(gpu, tr, case):
    """f16 operands at odd element offsets with odd leading dimensions and batch strides go straight into the MFMA kernels (LDS-DMA and 16-byte stores at
    element-aligned addresses; no padded copies since round 6): the same kernels on the same numbers, so the result must be the aligned call's, bit for bit --
    and nothing outside the output view may change."""
    wg = _wg()
    (M, N, K, mats) = case
    rng = np.random.default_rng(M + 3 * N + 7 * K + mats + int(tr))
    ar, ac = (K, M) if tr else (M, K)
    A = (rng.random((mats, ac, ar), dtype=np.float32) - 0.5).astype(np.float16)   # [matrix][column][row]
    B = (rng.random((mats, N, K), dtype=np.float32) - 0.5).astype(np.float16)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()

    def run(pads, offs, bpads):
        (pa, pb, pc), (oa, ob, oc), (ba, bb, bc) = pads, offs, bpads
        lda, ldb, ldc = ar + pa, K + pb, M + pc
        sa, sb, sc = lda * ac + ba, ldb * N + bb, ldc * N + bc  # batch strides
        fa = np.zeros(oa + sa * mats + 8, np.float16); fb = np.zeros(ob + sb * mats + 8, np.float16)
        fc = rng.random(oc + sc * mats + 8, dtype=np.float32).astype(np.float16)
        for z in range(mats):
            fa[oa + z * sa: oa + z * sa + lda * ac].reshape(ac, lda)[:, :ar] = A[z]
            fb[ob + z * sb: ob + z * sb + ldb * N].reshape(N, ldb)[:, :K] = B[z]
        ta, tb, tc = upload(gpu, (fa.size,), fa, np.float16), upload(gpu, (fb.size,), fb, np.float16), upload(gpu, (fc.size,), fc, np.float16)
        av = wg.GpuTensorView(wg.ViewShape((ar, ac, mats), lda, sa, oa), ta, 2)
        bv = wg.GpuTensorView(wg.ViewShape((K, N, mats), ldb, sb, ob), tb, 2)
        cv = wg.GpuTensorView(wg.ViewShape((M, N, mats), ldc, sc, oc), tc, 2)
        run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, cv, av, bv, variant))
        got = tc.read(gpu.device())
        out = np.stack([got[oc + z * sc: oc + z * sc + ldc * N].reshape(N, ldc)[:, :M] for z in range(mats)])
        mask = np.ones(fc.size, bool)
        for z in range(mats):
            mask[(oc + z * sc + np.arange(M)[None, :] + np.arange(N)[:, None] * ldc).ravel()] = False
        assert np.array_equal(got[mask].view(np.uint16), fc[mask].view(np.uint16)), "wrote outside the output view"
        return out

    want = run((0, 0, 0), (0, 0, 0), (0, 0, 0))
    ref = np.stack([(A[z].astype(np.float64).T if not tr else A[z].astype(np.float64)) @ B[z].astype(np.float64).T for z in range(mats)])  # [z][M][N]
    assert np.abs(want.transpose(0, 2, 1).astype(np.float64) - ref).max() <= 2.0 ** -10 * max(1.0, np.abs(ref).max()) + K * 2.0 ** -22
    for pads, offs, bpads in [((1, 3, 5), (1, 3, 5), (1, 1, 1)), ((4, 0, 0), (0, 4, 0), (0, 0, 4)), ((0, 0, 7), (2, 0, 1), (0, 3, 0)), ((8, 8, 8), (7, 6, 5), (2, 2, 2))]:
        got = run(pads, offs, bpads)
        if tr and N <= 16:  # the few-column streaming kernel (gemm_f32_skinny.hip, T = f16) keeps its 16-byte contract: off it, the tiled kernels -- another order of the sums
            assert np.abs(got.transpose(0, 2, 1).astype(np.float64) - ref).max() <= 2.0 ** -10 * max(1.0, np.abs(ref).max()) + K * 2.0 ** -22
            continue
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), (case, tr, pads, offs, bpads)


def test_copy_view_errors_and_skips(gpu):
    """wg_copy_view: a batch-count mismatch is the reference's kind of panic, a view past its buffer is an error, zero-sized views are skipped, an empty source zero-fills."""
    wg = _wg()
    from wgmath_amd import _lib
    a = np.arange(64, dtype=np.float32)
    ta, tb = upload(gpu, (64,), a), upload(gpu, (64,), np.full(64, 7.0, np.float32))
    dt = wg.wgcore.wg_dtype(np.float32)
    call = lambda dv, sv: _lib.check(_lib.lib.wg_copy_view(gpu._ctx.handle, dt, tb._h, dv.to_c(), ta._h, sv.to_c()))
    with pytest.raises(wg.DimensionMismatch, match="CopyView: dimension mismatch"):
        call(wg.ViewShape((4, 4, 2), 4, 16, 0), wg.ViewShape((4, 4, 1), 4, 16, 0))
    with pytest.raises(_lib.WgError, match="CopyView"):
        call(wg.ViewShape((4, 4, 1), 4, 16, 60), wg.ViewShape((4, 4, 1), 4, 16, 0))   # dst view runs past its buffer
    with pytest.raises(_lib.WgError, match="CopyView"):
        call(wg.ViewShape((4, 4, 1), 4, 16, 0), wg.ViewShape((4, 4, 1), 4, 16, 61))   # src view runs past its buffer
    call(wg.ViewShape((0, 4, 1), 4, 16, 0), wg.ViewShape((4, 4, 1), 4, 16, 0))        # nothing to write
    assert np.array_equal(tb.read(gpu.device()), np.full(64, 7.0, np.float32))
    call(wg.ViewShape((3, 2, 1), 5, 10, 1), wg.ViewShape((0, 0, 1), 1, 0, 0))         # an empty source: zeros
    want = np.full(64, 7.0, np.float32)
    want[1:4] = 0; want[6:9] = 0
    assert np.array_equal(tb.read(gpu.device()), want)
    # the operator form of the mirror (wgmath_amd.CopyView): a 5 x 3 block at offset 3 of `a` into offset 2 of `b`
    dst = wg.GpuTensorView(wg.ViewShape((5, 3, 1), 6, 18, 2), tb, 2)
    src = wg.GpuTensorView(wg.ViewShape((5, 3, 1), 7, 21, 3), ta, 2)
    cv, shapes = wg.CopyView.from_device(gpu.device()), wg.ViewShapeBuffers()
    run_pass(gpu, lambda p: cv.dispatch(gpu.device(), shapes, p, dst, src))
    for j in range(3):
        want[2 + 6 * j: 2 + 6 * j + 5] = a[3 + 7 * j: 3 + 7 * j + 5]
    assert np.array_equal(tb.read(gpu.device()), want)


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("MK", [(512, 512), (4096, 1024), (256, 8192), (8192, 256), (2048, 328), (1024, 1024)])
def test_gemm_f16_any_number_of_columns(gpu, tr, MK):
    """f16 Gemm / GemmTr onto N columns that are not a multiple of 4 run on the kernels as they are (no padded copies of m2 and out since round 6): every
    kernel family the launcher picks over N = 9 .. 4097, against f64, and not one element outside the M x N view written (the output's leading dimension
    leaves a gap after every column and the buffer goes on behind the last one)."""
    wg = _wg()
    M, K = MK
    rng = np.random.default_rng(M + 7 * K + int(tr))
    ar, ac = (K, M) if tr else (M, K)
    A = (rng.random((ac, ar), dtype=np.float32) - 0.5).astype(np.float16)
    ta = upload(gpu, (A.size,), A.ravel(), np.float16)
    av = wg.GpuTensorView(wg.ViewShape((ar, ac, 1), ar, ar * ac, 0), ta, 2)
    A64 = A.astype(np.float64).T if not tr else A.astype(np.float64)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    for N in (9, 13, 17, 31, 33, 63, 65, 127, 129, 255, 257, 1001, 4097):
        if M * N * K > (1 << 33):
            continue
        B = (rng.random((N, K), dtype=np.float32) - 0.5).astype(np.float16)
        ldc = M + 4
        fc = rng.random(ldc * N + 64, dtype=np.float32).astype(np.float16)
        tb, tc = upload(gpu, (B.size,), B.ravel(), np.float16), upload(gpu, (fc.size,), fc, np.float16)
        bv = wg.GpuTensorView(wg.ViewShape((K, N, 1), K, K * N, 0), tb, 2)
        cv = wg.GpuTensorView(wg.ViewShape((M, N, 1), ldc, ldc * N, 0), tc, 2)
        run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, cv, av, bv, variant))
        got = tc.read(gpu.device())
        ref = A64 @ B.astype(np.float64).T
        idx = np.arange(M)[:, None] + np.arange(N)[None, :] * ldc
        G = got[idx].astype(np.float64)
        assert np.abs(G - ref).max() <= 2.0 ** -10 * max(1.0, np.abs(ref).max()) + K * 2.0 ** -22, (M, K, N, tr)
        mask = np.ones(fc.size, bool)
        mask[idx.ravel()] = False
        assert np.array_equal(got[mask].view(np.uint16), fc[mask].view(np.uint16)), f"wrote outside the output view (M {M} K {K} N {N})"


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("MK", [(512, 512), (4096, 256), (256, 8192), (64, 4096), (96, 512), (2048, 132), (1024, 1024), (16, 2048), (128, 1024), (132, 1024), (1536, 192), (192, 4096)])
def test_gemm_f32_any_number_of_columns(gpu, tr, MK):
    """The f32 twin of the test above: N = 9 .. 4097 over the f32 launcher's families (main kernel, tail split, mid-size tiles, few rows / few columns, split-K).
    From 129 rows on f32 takes these N as they are too; up to 128 rows it keeps the zero-padded copies of m2 and out (the few-row forms compute the transposed
    product, whose row count N then is: without the copies 5 of 18 cases here were wrong)."""
    wg = _wg()
    M, K = MK
    rng = np.random.default_rng(3 * M + 7 * K + int(tr))
    ar, ac = (K, M) if tr else (M, K)
    A = (rng.random((ac, ar), dtype=np.float32) - 0.5)
    ta = upload(gpu, (A.size,), A.ravel())
    av = wg.GpuTensorView(wg.ViewShape((ar, ac, 1), ar, ar * ac, 0), ta, 2)
    A64 = A.astype(np.float64).T if not tr else A.astype(np.float64)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    for N in (9, 13, 17, 31, 33, 63, 65, 127, 129, 255, 257, 1001, 4097):
        if M * N * K > (1 << 32):
            continue
        B = (rng.random((N, K), dtype=np.float32) - 0.5)
        ldc = M + 4
        fc = rng.random(ldc * N + 64, dtype=np.float32)
        tb, tc = upload(gpu, (B.size,), B.ravel()), upload(gpu, (fc.size,), fc)
        bv = wg.GpuTensorView(wg.ViewShape((K, N, 1), K, K * N, 0), tb, 2)
        cv = wg.GpuTensorView(wg.ViewShape((M, N, 1), ldc, ldc * N, 0), tc, 2)
        run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, cv, av, bv, variant))
        got = tc.read(gpu.device())
        B64 = B.astype(np.float64).T
        ref, sabs = A64 @ B64, np.abs(A64) @ np.abs(B64)
        idx = np.arange(M)[:, None] + np.arange(N)[None, :] * ldc
        G = got[idx].astype(np.float64)
        tol = U.f32_gate(K, sabs)
        assert (np.abs(G - ref) <= tol).all(), f"M {M} K {K} N {N} tr {tr}: worst err/tol {(np.abs(G - ref) / tol).max():.3g}"
        mask = np.ones(fc.size, bool)
        mask[idx.ravel()] = False
        assert np.array_equal(got[mask].view(np.uint32), fc[mask].view(np.uint32)), f"wrote outside the output view (M {M} K {K} N {N})"


F32_ANY_ALIGN = [
    # M, N, K, matrices -- the f32 kernel families (gemm_f32.hip launcher): the 256 x 128 kernel (DMA interior + edge tiles), its tail split, the mid-size tiles, K split over the
    # workgroup's waves, few columns / few rows (gemm_f32_skinny.hip, 16- and 32-wide), split-K slabs, a batch, the 64 < M <= 128 transposed form
    (1536, 1280, 192, 1), (2304, 2048, 256, 1), (96, 96, 512, 1), (64, 64, 4096, 1), (2048, 16, 1024, 1), (2048, 48, 512, 1), (16, 2048, 1024, 1), (128, 128, 8192, 1),
    (1024, 768, 256, 3), (96, 8192, 256, 1),
]


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("case", F32_ANY_ALIGN)
def test_gemm_f32_any_offset_and_leading_dimension(gpu, tr, case):
    """f32 operands at odd element offsets with odd leading dimensions and batch strides run on the tuned kernels as they are (round 6: LDS-DMA and 16-byte
    accesses at element-aligned addresses; padded copies only for lengths): against f64 with the usual bound (the launcher's tile choice looks at leading
    dimensions, so the aligned call need not be the same kernel), and nothing outside the output view may change."""
    wg = _wg()
    (M, N, K, mats) = case
    rng = np.random.default_rng(11 * M + 3 * N + 7 * K + mats + int(tr))
    ar, ac = (K, M) if tr else (M, K)
    A = (rng.random((mats, ac, ar), dtype=np.float32) - 0.5)
    B = (rng.random((mats, N, K), dtype=np.float32) - 0.5)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    A64 = [(A[z].astype(np.float64).T if not tr else A[z].astype(np.float64)) for z in range(mats)]
    ref = np.stack([A64[z] @ B[z].astype(np.float64).T for z in range(mats)])
    sabs = np.stack([np.abs(A64[z]) @ np.abs(B[z].astype(np.float64).T) for z in range(mats)])
    for (pa, pb, pc), (oa, ob, oc), (ba, bb, bc) in [((0, 0, 0), (0, 0, 0), (0, 0, 0)), ((1, 3, 5), (1, 3, 5), (1, 1, 1)), ((4, 0, 0), (0, 2, 0), (0, 0, 3)), ((7, 2, 1), (3, 0, 2), (0, 3, 0))]:
        lda, ldb, ldc = ar + pa, K + pb, M + pc
        sa, sb, sc = lda * ac + ba, ldb * N + bb, ldc * N + bc
        fa = np.zeros(oa + sa * mats + 8, np.float32); fb = np.zeros(ob + sb * mats + 8, np.float32)
        fc = rng.random(oc + sc * mats + 8, dtype=np.float32)
        for z in range(mats):
            fa[oa + z * sa: oa + z * sa + lda * ac].reshape(ac, lda)[:, :ar] = A[z]
            fb[ob + z * sb: ob + z * sb + ldb * N].reshape(N, ldb)[:, :K] = B[z]
        ta, tb, tc = upload(gpu, (fa.size,), fa), upload(gpu, (fb.size,), fb), upload(gpu, (fc.size,), fc)
        av = wg.GpuTensorView(wg.ViewShape((ar, ac, mats), lda, sa, oa), ta, 2)
        bv = wg.GpuTensorView(wg.ViewShape((K, N, mats), ldb, sb, ob), tb, 2)
        cv = wg.GpuTensorView(wg.ViewShape((M, N, mats), ldc, sc, oc), tc, 2)
        run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, cv, av, bv, variant))
        got = tc.read(gpu.device())
        mask = np.ones(fc.size, bool)
        for z in range(mats):
            idx = oc + z * sc + np.arange(M)[:, None] + np.arange(N)[None, :] * ldc
            G = got[idx].astype(np.float64)
            tol = U.f32_gate(K, sabs[z])
            assert (np.abs(G - ref[z]) <= tol).all(), f"{case} {tr} pads {(pa, pb, pc)} offsets {(oa, ob, oc)}: worst err/tol {(np.abs(G - ref[z]) / tol).max():.3g}"
            mask[idx.ravel()] = False
        assert np.array_equal(got[mask].view(np.uint32), fc[mask].view(np.uint32)), "wrote outside the output view"


# One thing off at a time: only the operands that need a staged copy get one (api.hip gemm_staged / gemv_staged), so every combination of
# "as it lies" and "copied" operands must give the product -- and leave everything outside the output view alone.
GEMM_ONE_OFF = [
    # M, K, N, offsets (a, b, out), what is off
    (64, 32, 1, (0, 0, 0)), (64, 32, 2, (0, 0, 0)), (64, 32, 19, (0, 0, 0)),  # N only: m2 and out copied, m1 as it lies
    (64, 30, 16, (0, 0, 0)),                                                   # K only: m1 and m2 copied
    (61, 32, 16, (0, 0, 0)),                                                   # M only: m1 and out copied
    (64, 32, 16, (1, 0, 0)), (64, 32, 16, (0, 2, 0)), (64, 32, 16, (0, 0, 3)),  # one view at an odd offset
    (64, 32, 16, (0, 1, 1)), (516, 260, 3, (0, 0, 0)),
]


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("case", GEMM_ONE_OFF)
def test_gemm_one_operand_unaligned(gpu, dtype, tr, case):
    wg, wo = _wg(), _wo()
    (M, K, N, (oa, ob, oo)) = case
    rng = np.random.default_rng(M * 131 + K * 17 + N + oa + 2 * ob + 4 * oo + int(tr))
    ar, ac = (K, M) if tr else (M, K)
    lda, ldb, ldc = ar + 4, K + 8, M + 4  # (vec4-aligned leading dimensions: the offsets and sizes are what is off)
    pa = (rng.random(8 + lda * ac, dtype=np.float32) - 0.5).astype(dtype)
    pb = (rng.random(8 + ldb * N, dtype=np.float32) - 0.5).astype(dtype)
    po0 = rng.random(8 + ldc * N, dtype=np.float32).astype(dtype)
    ta, tb, to = upload(gpu, (pa.size,), pa, dtype), upload(gpu, (pb.size,), pb, dtype), upload(gpu, (po0.size,), po0, dtype)
    a_view = wg.GpuTensorView(wg.ViewShape((ar, ac, 1), lda, lda * ac, oa), ta, 2)
    b_view = wg.GpuTensorView(wg.ViewShape((K, N, 1), ldb, ldb * N, ob), tb, 2)
    o_view = wg.GpuTensorView(wg.ViewShape((M, N, 1), ldc, ldc * N, oo), to, 2)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, o_view, a_view, b_view, variant))
    got = to.read(gpu.device())
    sh = lambda v: wo.Shape(v.shape().size[0], v.shape().size[1], v.shape().size[2], v.shape().stride, v.shape().stride_mat, v.shape().offset)
    A, B = wo.view(pa, sh(a_view))[:, :, 0].astype(np.float64), wo.view(pb, sh(b_view))[:, :, 0].astype(np.float64)
    A = A.T if tr else A
    truth, sabs = A @ B, np.abs(A) @ np.abs(B)
    G = wo.view(got, sh(o_view))[:, :, 0].astype(np.float64)
    tol = U.f32_gate(K, sabs) + (2.0 ** -11 * np.abs(truth) + 2.0 ** -25 if dtype == np.float16 else 0.0)
    assert (np.abs(G - truth) <= tol).all(), f"gemm {case}: worst err/tol {(np.abs(G - truth) / tol).max():.3g}"
    mask = np.ones(po0.size, bool)
    mask[(oo + np.arange(M)[:, None] + np.arange(N)[None, :] * ldc).ravel()] = False
    assert np.array_equal(got[mask], po0[mask]), "wrote outside the output view"
    # beta on an output that is used where it lies / copied: out = A B + out
    from wgmath_amd import _lib
    to2 = upload(gpu, (po0.size,), po0, dtype)
    _lib.check(_lib.lib.wg_gemm_ex(gpu._ctx.handle, int(variant), wg.wgcore.wg_dtype(dtype), 1.0, 1.0, to2._h, o_view.shape().to_c(), ta._h, a_view.shape().to_c(),
                                   tb._h, b_view.shape().to_c()))
    got2 = wo.view(to2.read(gpu.device()), sh(o_view))[:, :, 0].astype(np.float64)
    C0 = wo.view(po0, sh(o_view))[:, :, 0].astype(np.float64)
    tol2 = tol + 4 * 2.0 ** (-11 if dtype == np.float16 else -24) * (np.abs(truth + C0) + np.abs(C0)) + 1e-6
    assert (np.abs(got2 - (truth + C0)) <= tol2).all()


GEMV_ONE_OFF = [
    # R, C, nrhs, offsets (m, v, out)
    (64, 32, 1, (0, 1, 0)), (64, 32, 1, (0, 0, 2)), (64, 32, 3, (1, 0, 0)), (61, 32, 1, (0, 0, 0)), (64, 30, 2, (0, 0, 0)), (516, 260, 1, (0, 3, 3)),
]


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("case", GEMV_ONE_OFF)
def test_gemv_one_operand_unaligned(gpu, tr, case):
    wg, wo = _wg(), _wo()
    (R, C, n, (om, ov, oo)) = case
    rng = np.random.default_rng(R * 7 + C * 3 + n + om + 2 * ov + 4 * oo + int(tr))
    ldm = R + 4
    vlen, olen = (R, C) if tr else (C, R)
    ldv, ldo = vlen + 4 - vlen % 4, olen + 4 - olen % 4
    pm = (rng.random(8 + ldm * C, dtype=np.float32) - 0.5).astype(np.float32)
    pv = (rng.random(8 + ldv * n, dtype=np.float32) - 0.5).astype(np.float32)
    po = rng.random(8 + ldo * n, dtype=np.float32)
    tm, tv, to = upload(gpu, (pm.size,), pm), upload(gpu, (pv.size,), pv), upload(gpu, (po.size,), po)
    m_view = wg.GpuTensorView(wg.ViewShape((R, C, 1), ldm, ldm * C, om), tm, 2)
    v_view = wg.GpuTensorView(wg.ViewShape((vlen, n, 1), ldv, ldv * n, ov), tv, 2)
    o_view = wg.GpuTensorView(wg.ViewShape((olen, n, 1), ldo, ldo * n, oo), to, 2)
    variant = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv
    gemv, shapes = wg.Gemv.from_device(gpu.device()), wg.ViewShapeBuffers()
    run_pass(gpu, lambda p: gemv.dispatch_generic(gpu.device(), shapes, p, o_view, m_view, v_view, variant))
    got = to.read(gpu.device())
    sh = lambda v: wo.Shape(v.shape().size[0], v.shape().size[1], v.shape().size[2], v.shape().stride, v.shape().stride_mat, v.shape().offset)
    A, X = wo.view(pm, sh(m_view))[:, :, 0], wo.view(pv, sh(v_view))[:, :, 0]
    A = A.T if tr else A
    truth, sabs = wo.gemm_f64(A, X)
    U.assert_close_f64(wo.view(got, sh(o_view))[:, :, 0], truth, vlen, sabs, f"gemv {case} vs f64")
    mask = np.ones(po.size, bool)
    mask[(oo + np.arange(olen)[:, None] + np.arange(n)[None, :] * ldo).ravel()] = False
    assert np.array_equal(got[mask], po[mask]), "wrote outside the output view"


GEMV_ANY = [
    # R, C, ld, offsets (m, v, out), nrhs, mats -- matrix views the vec4 kernels cannot address: one pass where they lie (gemv_any.hip)
    (1000, 777, 1001, (3, 1, 2), 1, 1), (4097, 300, 4099, (1, 0, 0), 2, 1), (64, 5000, 66, (2, 3, 1), 1, 2), (8193, 64, 8193, (0, 0, 0), 1, 1),
    (5, 3, 5, (1, 0, 0), 1, 1), (513, 513, 515, (7, 5, 3), 3, 2), (20000, 16, 20002, (2, 0, 1), 1, 1), (16, 20000, 18, (1, 1, 1), 1, 1), (2048, 2048, 2048, (1, 0, 0), 1, 1),
    # many whole row blocks + a ragged one, column / row splits with partials, an odd leading dimension (every column its own alignment)
    (8193, 4099, 8195, (1, 0, 1), 1, 1), (4099, 8193, 4101, (3, 2, 0), 2, 1),
]


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("case", GEMV_ANY)
def test_gemv_any_alignment(gpu, dtype, tr, case):
    """Against f64; an Inf in the element just past every column's end (the neighbour in the parent buffer) must not leak in; nothing outside the output view is touched."""
    wg, wo = _wg(), _wo()
    (R, C, ld, (om, ov, oo), n, mats) = case
    rng = np.random.default_rng(R * 3 + C + ld + om + int(tr))
    vlen, olen = (R, C) if tr else (C, R)
    pm = (rng.random(om + ld * C * mats + 8, dtype=np.float32) - 0.5).astype(dtype)
    if ld > R:  # what follows a column in memory belongs to the parent, not to the view
        for zc in range(C * mats):
            pm[om + zc * ld + R] = np.inf
    pv = (rng.random(ov + (vlen + 1) * n * mats + 8, dtype=np.float32) - 0.5).astype(dtype)
    po = rng.random(oo + (olen + 3) * n * mats + 8, dtype=np.float32).astype(dtype)
    tm, tv, to = upload(gpu, (pm.size,), pm, dtype), upload(gpu, (pv.size,), pv, dtype), upload(gpu, (po.size,), po, dtype)
    m_view = wg.GpuTensorView(wg.ViewShape((R, C, mats), ld, ld * C, om), tm, 2)
    v_view = wg.GpuTensorView(wg.ViewShape((vlen, n, mats), vlen + 1, (vlen + 1) * n, ov), tv, 2)
    o_view = wg.GpuTensorView(wg.ViewShape((olen, n, mats), olen + 3, (olen + 3) * n, oo), to, 2)
    variant = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv
    gemv, shapes = wg.Gemv.from_device(gpu.device()), wg.ViewShapeBuffers()
    run_pass(gpu, lambda p: gemv.dispatch_generic(gpu.device(), shapes, p, o_view, m_view, v_view, variant))
    got = to.read(gpu.device())
    mask = np.ones(po.size, bool)
    for z in range(mats):
        A = pm[om + z * ld * C: om + z * ld * C + ld * C].reshape(C, ld)[:, :R].T.astype(np.float64)
        X = pv[ov + z * (vlen + 1) * n: ov + (z + 1) * (vlen + 1) * n].reshape(n, vlen + 1)[:, :vlen].T.astype(np.float64)
        A = A.T if tr else A
        truth, sabs = A @ X, np.abs(A) @ np.abs(X)
        idx = oo + z * (olen + 3) * n + np.arange(olen)[:, None] + np.arange(n)[None, :] * (olen + 3)
        G = got[idx].astype(np.float64)
        tol = U.f32_gate(vlen, sabs) + (2.0 ** -11 * np.abs(truth) + 2.0 ** -25 if dtype == np.float16 else 0.0)
        assert np.isfinite(G).all() and (np.abs(G - truth) <= tol).all(), f"gemv {case}: worst err/tol {(np.abs(G - truth) / tol).max():.3g}"
        mask[idx.ravel()] = False
    assert np.array_equal(got[mask].view(np.uint8), po[mask].view(np.uint8)), "wrote outside the output view"


@pytest.mark.parametrize("tr", [False, True])
def test_gemv_unaligned_views(gpu, tr):
    """Gemv on a matrix view at an odd row / with odd lengths and vectors at odd offsets (GpuVector::rows(1, n), tensor.rs:669-680), 3
    right-hand sides: the any-alignment kernels, checked against f64; nothing outside the output view touched."""
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(654 + int(tr))
    PR, PC = 203, 150
    pm = (rng.random(PR * PC, dtype=np.float32) - 0.5).astype(np.float32)
    tm = upload(gpu, (PR, PC), pm)
    R, C = 131, 94
    m_view = wg.GpuTensorView(wg.ViewShape((R, C, 1), PR, PR * PC, 3 + 7 * PR), tm, 2)
    vlen, olen = (R, C) if tr else (C, R)
    pv = (rng.random(5 + 3 * (vlen + 3), dtype=np.float32) - 0.5).astype(np.float32)
    po = rng.random(7 + 3 * (olen + 1), dtype=np.float32)
    tv, to = upload(gpu, (pv.size,), pv), upload(gpu, (po.size,), po)
    v_view = wg.GpuTensorView(wg.ViewShape((vlen, 3, 1), vlen + 3, 1, 5), tv, 2)
    o_view = wg.GpuTensorView(wg.ViewShape((olen, 3, 1), olen + 1, 1, 7), to, 2)
    variant = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv
    gemv, shapes = wg.Gemv.from_device(gpu.device()), wg.ViewShapeBuffers()
    run_pass(gpu, lambda p: gemv.dispatch_generic(gpu.device(), shapes, p, o_view, m_view, v_view, variant))
    got = to.read(gpu.device())
    sh = lambda v: wo.Shape(v.shape().size[0], v.shape().size[1], v.shape().size[2], v.shape().stride, v.shape().stride_mat, v.shape().offset)
    A, X = wo.view(pm, sh(m_view))[:, :, 0], wo.view(pv, sh(v_view))[:, :, 0]
    A = A.T if tr else A
    truth, sabs = wo.gemm_f64(A, X)
    U.assert_close_f64(wo.view(got, sh(o_view))[:, :, 0], truth, vlen, sabs, "unaligned gemv vs f64")
    mask = np.ones(po.size, bool)
    s_ = sh(o_view).resolved()
    mask[(s_.offset + np.arange(olen)[:, None] + np.arange(3)[None, :] * s_.stride).ravel()] = False
    assert np.array_equal(got[mask], po[mask]), "unaligned gemv wrote outside its output view"


GEMV_SHAPES = [
    # R,    C, nrhs, mats
    (4, 4, 1, 1), (8, 260, 1, 1), (260, 8, 1, 1), (1024, 1024, 1, 1), (512, 4096, 2, 1), (4096, 512, 3, 2),
    (252, 1000, 5, 1), (128, 65536, 1, 1), (65536, 128, 1, 1), (2048, 2048, 4, 1),
    # RHS register tiles 8 (5..8 columns) and more than one RHS group (> 8 columns), with a batch
    (1024, 2048, 8, 1), (516, 772, 7, 2), (256, 1024, 9, 1), (1024, 512, 12, 2), (64, 128, 17, 1),
    # 9 .. 64 right-hand sides with >= 512 outputs and k >= 128: one pass on the few-column Gemm kernels
    (1024, 512, 12, 1), (1024, 516, 17, 2), (516, 1028, 40, 1), (1024, 640, 64, 1),
    # GemvTr with one right-hand side (a half-wave per column): batches, ragged lengths, splits, the 64-KiB-column case of the older kernel
    (12, 12, 1, 3), (1028, 36, 1, 2), (16384, 64, 1, 1), (40004, 68, 1, 1), (300000, 4, 1, 1), (2052, 4100, 1, 1),
    # 3 .. 8 right-hand sides on matrices past the launch-bound sizes: one pass on the matrix cores (few_rhs_as_gemm in gemv.hip)
    (4096, 4096, 4, 1), (4096, 3072, 8, 1), (2048, 2304, 7, 1),
    # GemvTr, 2 .. 8 right-hand sides, >= 128 x CUs outputs, vectors in the LDS (gemv_t_lds_kernel); the transposed shape is the plain multi-vector Gemv
    (64, 32768, 3, 1), (132, 33000, 8, 1), (260, 32772, 2, 2),
    # ... two right-hand sides from 8 outputs per CU on (round 5: the kernel's narrow workgroup shapes)
    (2048, 6004, 2, 1), (1024, 4096, 2, 2),
    # 2 .. 8 right-hand sides on more shapes (round 5): few outputs with a long contraction, a ragged column count with a batch, k off every tile size
    (65536, 128, 3, 1), (8192, 1028, 5, 2), (4100, 2048, 4, 1), (16384, 512, 8, 1), (6000, 3000, 2, 1),
    # GemvTr, two right-hand sides on the half-wave-per-column kernel's 2-vector form (round 5: contractions of 2049 .. 8192 rows onto 4096 .. 16384 outputs): whole trips, a
    # partial trip + ragged rows, a batch
    (4096, 4100, 2, 1), (3000, 8192, 2, 2), (8192, 4096, 2, 1),
]
# gemv_t_lds_kernel forced (WG_TUNE_GEMVT_LDS = 8 outputs per CU) onto shapes the launcher keeps on other kernels: every workgroup shape (8 / 16 / 32 / 64 / 128
# columns per trip: gemv_t_lds_plan), vectors longer than the LDS in 2 - 4 chunks (8 right-hand sides x 8192 rows = 256 KiB; a ragged last chunk; a batch), a
# last column group that is not full
GEMVT_LDS_SHAPES = [(512, 2048, 3, 1), (1024, 4096, 4, 1), (4096, 11008, 4, 1), (256, 16384, 5, 1), (8192, 2048, 8, 1), (12288, 2056, 4, 1), (9000, 2052, 3, 2),
                    (2048, 6004, 2, 1), (128, 40000, 7, 1)]


@pytest.mark.parametrize("R,C,nrhs,mats", GEMVT_LDS_SHAPES)
def test_gemv_tr_lds_kernel_every_shape(gpu, oracle_c, R, C, nrhs, mats):
    old = gpu.set_tuning("gemvt_lds", 8)
    try:
        test_gemv_shapes(gpu, oracle_c, R, C, nrhs, mats, True)
    finally:
        gpu.set_tuning("gemvt_lds", old)



@pytest.mark.parametrize("R,C,nrhs,mats", GEMV_SHAPES)
@pytest.mark.parametrize("tr", [False, True])
def test_gemv_shapes(gpu, oracle_c, R, C, nrhs, mats, tr):
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(R * 7919 + C * 31 + nrhs * 5 + mats + int(tr))
    m = (rng.random(R * C * mats, dtype=np.float32) * 2 - 1).astype(np.float32)
    vlen, olen = (R, C) if tr else (C, R)
    v = (rng.random(vlen * nrhs * mats, dtype=np.float32) * 2 - 1).astype(np.float32)
    o0 = rng.random(olen * nrhs * mats, dtype=np.float32)
    sm, sv, so = wo.Shape(R, C, mats), wo.Shape(vlen, nrhs, mats), wo.Shape(olen, nrhs, mats)
    variant = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv
    orc = o0.copy()
    oracle_c.gemv(int(variant), orc, so, m, sm, v, sv)
    tm, tv, to = upload(gpu, (R, C, mats), m), upload(gpu, (vlen, nrhs, mats), v), upload(gpu, (olen, nrhs, mats), o0)
    gemv, shapes = wg.Gemv.from_device(gpu.device()), wg.ViewShapeBuffers()
    run_pass(gpu, lambda p: gemv.dispatch_generic(gpu.device(), shapes, p, to, tm, tv, variant))
    got = to.read(gpu.device())
    A, X = wo.view(m, sm), wo.view(v, sv)
    for t in range(mats):
        amk = A[:, :, t].T if tr else A[:, :, t]
        truth, sabs = wo.gemm_f64(amk, X[:, :, t])
        U.assert_close_f64(wo.view(got, so)[:, :, t], truth, vlen, sabs, f"gemv {R}x{C} rhs={nrhs} mat {t} tr={tr} vs f64")
        U.assert_close_oracle(wo.view(got, so)[:, :, t], wo.view(orc, so)[:, :, t], vlen, sabs, "vs oracle")


@pytest.mark.parametrize("n,off_a,off_b", [(1, 0, 0), (3, 1, 1), (1757, 0, 0), (1757, 3, 3), (1757, 1, 2), (100003, 5, 9), (1 << 20, 0, 4)])
@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_op_assign_offsets(gpu, n, off_a, off_b, dtype):
    wg = _wg()
    rng = np.random.default_rng(n + off_a * 17 + off_b)
    pa = (rng.random(n + 16, dtype=np.float32) * 4 - 2).astype(dtype)
    pb = (rng.random(n + 16, dtype=np.float32) * 4 - 2).astype(dtype)
    shapes = wg.ViewShapeBuffers()
    for op in wg.OpAssignVariant:
        ta, tb = upload(gpu, (n + 16,), pa, dtype), upload(gpu, (n + 16,), pb, dtype)
        oa = wg.OpAssign.new(gpu.device(), op)
        va, vb = ta.rows(off_a, n), tb.rows(off_b, n)
        run_pass(gpu, lambda p: oa.dispatch(gpu.device(), shapes, p, va, vb))
        got = ta.read(gpu.device())
        exp = pa.copy()
        x, y = pa[off_a:off_a + n].astype(np.float32), pb[off_b:off_b + n].astype(np.float32)
        with np.errstate(all="ignore"):
            r = {0: x + y, 1: x - y, 2: x * y, 3: x / y, 4: y}[int(op)]
        exp[off_a:off_a + n] = r.astype(dtype)
        U.assert_bits_equal(got, exp, f"op_assign {op!r} n={n} offsets ({off_a},{off_b}) {np.dtype(dtype).name}")


# --------------------------------------------------------------------------------------------------------
# error behaviour (what the reference turns into a panic / a silent skip)
# --------------------------------------------------------------------------------------------------------
def test_errors_and_skips(gpu):
    wg = _wg()
    dev, shapes = gpu.device(), wg.ViewShapeBuffers()
    z = lambda *shape: upload(gpu, shape, np.zeros(int(np.prod(shape)), np.float32))
    gemm, gemv = wg.Gemm.from_device(dev), wg.Gemv.from_device(dev)
    enc = dev.create_command_encoder()
    p = enc.compute_pass("errors", None)
    with pytest.raises(wg.DimensionMismatch, match="Gemm: dimension mismatch."):
        gemm.dispatch(dev, shapes, p, z(8, 8), z(8, 12), z(8, 8))
    with pytest.raises(AssertionError):  # DimensionMismatch is an AssertionError, like a Rust assert_eq! panic
        gemm.dispatch_tr(dev, shapes, p, z(8, 8), z(8, 12), z(8, 8))
    with pytest.raises(wg.DimensionMismatch, match="Gemv: dimension mismatch."):
        gemv.dispatch(dev, shapes, p, z(8), z(8, 12), z(8))
    with pytest.raises(wg.DimensionMismatch, match="Op-assign: dimension mismatch."):
        wg.OpAssign.new(dev, wg.OpAssignVariant.Add).dispatch(dev, shapes, p, z(8), z(12))
    with pytest.raises(wg.PreconditionFailed):  # gemv.rs:122
        gemv.dispatch_generic(dev, shapes, p, z(6), z(6, 8), z(8), wg.GemvVariant.GemvFast)
    gemm.dispatch(dev, shapes, p, z(6, 8), z(6, 8), z(8, 8))  # not vec4-aligned: computed all the same (test_gemm_unaligned_views), no error
    with pytest.raises(wg.WgError, match="addresses"):  # view larger than its buffer
        big = z(8, 8)
        gemm.dispatch(dev, shapes, p, wg.GpuTensorView(wg.ViewShape((16, 16, 1), 16, 256, 0), big, 2), z(16, 16), z(16, 16))
    # GemvTrFast with rows % 128 != 0 silently runs GemvTr (gemv.rs:99-104)
    m, v, o = upload(gpu, (8, 12), np.ones(96, np.float32)), upload(gpu, (8,), np.ones(8, np.float32)), z(12)
    gemv.dispatch_generic(dev, shapes, p, o, m, v, wg.GemvVariant.GemvTrFast)
    # zero-sized tensors: silently skipped (kernel.rs:111-123)
    gemm.dispatch(dev, shapes, p, z(0, 0), z(0, 0), z(0, 0))
    wg.OpAssign.new(dev, wg.OpAssignVariant.Add).dispatch(dev, shapes, p, z(0), z(0))
    p.end()
    gpu.queue().submit([enc.finish()])
    assert np.array_equal(o.read(dev), np.full(12, 8, np.float32))


# --------------------------------------------------------------------------------------------------------
# runtime: record/replay, timestamps, staging copies
# --------------------------------------------------------------------------------------------------------
def test_record_replay_and_timestamps(gpu):
    wg = _wg()
    dev, shapes = gpu.device(), wg.ViewShapeBuffers()
    n = 4096
    a = upload(gpu, (n,), np.zeros(n, np.float32))
    b = upload(gpu, (n,), np.ones(n, np.float32))
    add = wg.OpAssign.new(dev, wg.OpAssignVariant.Add)
    enc = dev.create_command_encoder(record=True)
    with enc.compute_pass("recorded", None) as p:
        add.dispatch(dev, shapes, p, a, b)
        add.dispatch(dev, shapes, p, a, b)
    cb = enc.finish()
    assert np.array_equal(a.read(dev), np.zeros(n, np.float32))  # nothing ran while recording
    for _ in range(5):
        gpu.queue().submit([cb])
    assert np.array_equal(a.read(dev), np.full(n, 10, np.float32))

    ts = wg.GpuTimestamps.new(dev, 8)
    enc = dev.create_command_encoder()
    with enc.compute_pass("timed", ts) as p:
        for _ in range(10):
            add.dispatch(dev, shapes, p, a, b)
    ts.resolve(enc)
    gpu.queue().submit([enc.finish()])
    t = ts.wait_for_results_ms()
    assert len(t) == 2 and t[0] == 0.0 and 0.0 < t[1] < 1000.0
    assert np.array_equal(a.slow_read(gpu), np.full(n, 20, np.float32))

    # the rest of timestamps.rs: slots reserved first (next_query_indices: all or none), a pass that writes its reserved pair, explicit writes inside a pass,
    # raw values and their conversion, the async forms, clear
    ts.clear()
    assert ts.is_empty() and ts.query_set() is ts
    writes = ts.next_compute_pass_timestamp_writes()
    assert (writes.beginning_of_pass_write_index, writes.end_of_pass_write_index) == (0, 1) and ts.len() == 2
    enc = dev.create_command_encoder()
    with enc.compute_pass("timed by reserved slots", writes) as p:
        add.dispatch(dev, shapes, p, a, b)
        assert ts.write_next_timestamp(p) == 2        # a third timestamp inside the pass
        add.dispatch(dev, shapes, p, a, b)
        assert ts.write_timestamp_at(p, 5) and not ts.write_timestamp_at(p, 8)  # slot 5 is below the capacity (never allocated: len stays), slot 8 is not
    gpu.queue().submit([enc.finish()])
    assert ts.next_query_indices(6) is None and ts.len() == 3   # 3 + 6 > 8: nothing is taken
    assert ts.next_query_indices(5) == [3, 4, 5, 6, 7] and ts.next_query_index() is None
    ms = ts.wait_for_results_ms(dev, gpu.queue())
    assert len(ms) == 8 and ms[0] == 0.0 and 0.0 < ms[2] < ms[1] < 1000.0 and ms[2] <= ms[5] <= ms[1] and ms[3] == ms[4] == ms[6] == ms[7] == 0.0
    raw = ts.wait_for_results(dev)
    assert all(isinstance(x, int) for x in raw) and raw[1] > raw[2] > 0
    back = wg.GpuTimestamps.timestamps_to_ms(raw, gpu.queue().get_timestamp_period())
    assert all(abs(x - y) < 1e-5 for x, y in zip(back, ms))
    import asyncio
    assert asyncio.run(ts.wait_for_results_async(dev)) == raw and asyncio.run(ts.wait_for_results_ms_async(gpu.queue(), dev)) == ms
    g2 = wg.GpuInstance.without_gl()
    assert g2.device_arc() is g2.device() and wg.GpuInstance.with_backends("vulkan").adapter()["compute_units"] > 0


def test_objects_dropped_during_a_recording_are_freed_after_it(gpu):
    """A tensor, a command buffer or a whole GpuInstance whose last reference goes away WHILE the thread records (Python's cyclic GC picks its
    moment; a Rust drop at scope end) must not touch the HIP runtime then: hipFree / hipStreamSynchronize from the capturing thread are prohibited
    and invalidate the capture. The destroy is queued and runs when the recording ends (wg_encoder_finish) -- the reference keeps a dropped
    buffer alive until the submission that uses it retires. (Found as a flaky failure of the record / replay tests: the GC freed an earlier
    test's tensor in the middle of a recording.)"""
    import gc
    wg = _wg()
    dev, shapes = gpu.device(), wg.ViewShapeBuffers()
    n = 4096
    a, b = upload(gpu, (n,), np.zeros(n, np.float32)), upload(gpu, (n,), np.ones(n, np.float32))
    victims = [upload(gpu, (1 << 16,), np.zeros(1 << 16, np.float32)) for _ in range(4)]
    other = wg.GpuInstance.new(0)  # a second context of this thread, with a buffer and a finished recording of its own
    ob = wg.TensorBuilder.vector(n, wg.BufferUsages.STORAGE).build(other.device(), np.float32)
    oenc = other.device().create_command_encoder(record=True)
    ocb = oenc.finish()
    add = wg.OpAssign.new(dev, wg.OpAssignVariant.Add)
    enc = dev.create_command_encoder(record=True)
    with enc.compute_pass("recorded", None) as p:
        add.dispatch(dev, shapes, p, a, b)
        del victims, ob, ocb, oenc
        other.close()
        del other
        gc.collect()  # every destroy call lands inside the recording
        add.dispatch(dev, shapes, p, a, b)
    cb = enc.finish()
    for _ in range(3):
        gpu.queue().submit([cb])
    assert np.array_equal(a.read(dev), np.full(n, 6, np.float32))
    gpu.sync()


@pytest.mark.gpu
def test_a_tensor_the_recording_used_and_the_host_dropped_outlives_every_replay(gpu):
    """The operand of a recorded dispatch loses its last host reference DURING the recording. Its destroy is queued on the command buffer (not run at
    wg_encoder_finish): every later submit still reads valid memory -- wgpu keeps a dropped buffer alive until the submission using it retires -- and
    the memory is returned when the command buffer is destroyed. Checked by value (the replays keep adding the operand's 3s; fresh allocations made
    after finish() would be handed the freed block and overwrite it) and by the allocator's own accounting."""
    import gc
    wg = _wg()
    dev, shapes = gpu.device(), wg.ViewShapeBuffers()
    n = 1 << 22                                                   # 16 MiB: a block of its own in the HIP allocator
    a = upload(gpu, (n,), np.zeros(n, np.float32))
    b = upload(gpu, (n,), np.full(n, 3, np.float32))
    gpu.sync()
    add = wg.OpAssign.new(dev, wg.OpAssignVariant.Add)
    enc = dev.create_command_encoder(record=True)
    with enc.compute_pass("recorded", None) as p:
        add.dispatch(dev, shapes, p, a, b)
        del b
        gc.collect()                                              # wg_buf_destroy(b) arrives inside the recording
    cb = enc.finish()
    free_with_b = gpu.mem_info()[0]
    squatters = [upload(gpu, (n,), np.full(n, -1000, np.float32)) for _ in range(4)]  # would land on b's block had finish() freed it
    for _ in range(3):
        gpu.queue().submit([cb])
    assert np.array_equal(a.read(dev), np.full(n, 9, np.float32))
    del squatters
    gc.collect()
    gpu.sync()
    assert abs(gpu.mem_info()[0] - free_with_b) < (4 << 20)
    del cb
    gc.collect()                                                  # wg_cmdbuf_destroy: the queued wg_buf_destroy(b) runs now
    gpu.sync()
    assert gpu.mem_info()[0] >= free_with_b + n * 4 - (4 << 20), "b's 16 MiB must come back when the command buffer goes"


@pytest.mark.gpu
def test_a_recording_context_is_finished_or_destroyed_on_its_own_thread(gpu):
    """Thread-local capture: only the thread that began a recording can end it, and the deferred-destroy bookkeeping is that thread's. wg_encoder_finish and
    wg_ctx_destroy from another thread are refused (an error, not a silent no-op or a leaked counter); the recording thread then finishes normally and the result replays."""
    import threading
    wg = _wg()
    from wgmath_amd._lib import lib
    other = wg.GpuInstance.new(0)
    dev, shapes = other.device(), wg.ViewShapeBuffers()
    n = 4096
    S = wg.BufferUsages
    a = wg.TensorBuilder.vector(n, S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, np.zeros(n, np.float32))
    b = wg.TensorBuilder.vector(n, S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, np.ones(n, np.float32))
    add = wg.OpAssign.new(dev, wg.OpAssignVariant.Add)
    enc = dev.create_command_encoder(record=True)
    with enc.compute_pass("recorded", None) as p:
        add.dispatch(dev, shapes, p, a, b)
        rcs = {}

        def elsewhere():
            import ctypes
            out = ctypes.c_void_p()
            rcs["finish"] = lib.wg_encoder_finish(other._ctx.handle, ctypes.byref(out))
            rcs["destroy"] = lib.wg_ctx_destroy(other._ctx.handle)
        t = threading.Thread(target=elsewhere)
        t.start()
        t.join()
        assert rcs["finish"] != 0 and rcs["destroy"] != 0, rcs
        add.dispatch(dev, shapes, p, a, b)
    cb = enc.finish()
    other.queue().submit([cb])
    other.queue().submit([cb])
    assert np.array_equal(a.read(dev), np.full(n, 4, np.float32))
    del cb
    other.sync()


# --------------------------------------------------------------------------------------------------------
# f16 Gemm (extension: no reference kernel -- contract defined in DESIGN.md: f16 in, f32 accumulate, one RNE rounding)
# --------------------------------------------------------------------------------------------------------
F16_SHAPES = [
    # M,   K,   N, mats      (multiples of 256/64 -> MFMA fast path; anything else -> generic path)
    (256, 64, 256, 1), (512, 128, 256, 1), (256, 192, 768, 2), (1024, 1024, 512, 1),
    (36, 20, 28, 1), (260, 72, 132, 1), (256, 60, 256, 1), (64, 1024, 64, 3),
    # ragged edge tiles on the MFMA path (M, N multiples of 8 but not of 256; K multiple of 32)
    (8, 32, 8, 1), (264, 96, 520, 1), (1000, 256, 776, 2), (4096 + 8, 64, 256 - 8, 1), (248, 32, 4104, 1),
    # split-K (few tiles, long K), incl. a K that does not divide evenly and a batch
    (256, 8192, 256, 1), (512, 4096 + 32, 264, 1), (1024, 16384, 512, 2),
    # 16x16x32 kernel (K % 64 == 0, >= 3 stages): minimum, odd and even stage counts, several tiles per CU-less grid
    (512, 192, 512, 1), (512, 320, 512, 1), (768, 448, 264, 2), (2048, 2048, 1024, 1),
    # shapes / alignments the MFMA kernels do not take as they are (K % 32, M or N % 8 -- multiples of 4 as the operator demands), large
    # enough for the zero-padded staging path
    (1000, 1000, 1000, 1), (2052, 520, 264, 2), (260, 1024, 1004, 1), (512, 4104, 512, 1), (1004, 260, 516, 3),
    # N a multiple of 4 only (columns are independent: no staging), on every MFMA kernel family
    (512, 256, 260, 1), (256, 192, 1004, 2), (1024, 64, 4100, 1),
    # GemmTr with N <= 16 on matrices of >= 16 MiB: the few-column streaming kernel in f16 (gemm_f32_skinny.hip, T = _Float16; round 5) -- a K that ends in a partial
    # stage, a batch, rows that are not a multiple of a wave's 32, k split across workgroups + reduce (3 / 5 / 8 columns: test_gemv_f16's GemvTr cases)
    (4096, 2048, 8, 1), (8192, 1096, 16, 1), (2048, 4104, 4, 2), (4100, 2048, 12, 1), (1024, 16384, 8, 1),
    # tail split: 17 x 17 = 289 tiles on 256 CUs -> 256 tiles as they are + 33 tiles cut along K (full and ragged tiles)
    (4352, 1024, 4352, 1), (4104, 512, 4104, 1),
]


@pytest.fixture(params=["auto", "128", "256", "256128"])
def f16_tile(request, gpu):
    """The f16 launcher picks between the 256 x 256 kernels and the 128 x 128 kernel (mid-size outputs) by shape; the context's
    WG_TUNE_F16_TILE knob (wg_ctx_set_tuning) forces one family, so that every shape below exercises both."""
    old = gpu.set_tuning("f16_tile", 0 if request.param == "auto" else int(request.param))
    yield request.param
    gpu.set_tuning("f16_tile", old)


def f16_check(got, a64, b64, K, what):
    truth = a64 @ b64
    sabs = np.abs(a64) @ np.abs(b64)
    tol = U.f32_gate(K, sabs) + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25  # + half an f16 ulp (+ subnormal floor)
    err = np.abs(np.asarray(got, np.float64) - truth)
    assert (err <= tol).all(), f"{what}: worst err/tol = {(err / tol).max():.3g} at {np.unravel_index((err / tol).argmax(), err.shape)}"


@pytest.mark.parametrize("M,K,N,mats", F16_SHAPES)
@pytest.mark.parametrize("tr", [False, True])
def test_gemm_f16_shapes(gpu, f16_tile, M, K, N, mats, tr):
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(M * 31 + K * 17 + N + mats + int(tr))
    a = (rng.random(M * K * mats, dtype=np.float32) * 2 - 1).astype(np.float16)
    b = (rng.random(K * N * mats, dtype=np.float32) * 2 - 1).astype(np.float16)
    s1 = wo.Shape(K, M, mats) if tr else wo.Shape(M, K, mats)
    s2, so = wo.Shape(K, N, mats), wo.Shape(M, N, mats)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    m1 = upload(gpu, (K, M, mats) if tr else (M, K, mats), a, np.float16)
    m2 = upload(gpu, (K, N, mats), b, np.float16)
    out = upload(gpu, (M, N, mats), np.full(M * N * mats, np.nan, np.float16), np.float16)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, m1, m2, variant))
    got = out.read(gpu.device())
    assert got.dtype == np.float16
    A, B = wo.view(a, s1), wo.view(b, s2)
    for t in range(mats):
        amk = (A[:, :, t].T if tr else A[:, :, t]).astype(np.float64)
        f16_check(wo.view(got, so)[:, :, t], amk, B[:, :, t].astype(np.float64), K, f"f16 gemm {M}x{K}x{N} mat {t} tr={tr}")
    # SURVEY 8(c)'s f16 oracle: the f32 WGSL restatement (oracle/wgsl_oracle.c) on the exactly representable f16 operands. The MFMA
    # kernels accumulate the same exact products in f32 in another order and round once to f16: both sit within the f32 gate of the
    # truth, so |gpu - oracle| <= 2 x gate + half an f16 ulp of the result (+ the subnormal floor). Shapes the C oracle finishes in
    # seconds (and whose dimensions the WGSL kernels take: multiples of 4).
    if (tr or f16_tile == "auto") and M * K * N * mats <= (1 << 31):
        C = wo.CLib()
        orc = np.zeros(M * N * mats, np.float32)
        C.gemm(wo.GEMM_TR if tr else wo.GEMM, orc, so, a.astype(np.float32), s1, b.astype(np.float32), s2)
        O = wo.view(orc, so)
        for t in range(mats):
            amk = (A[:, :, t].T if tr else A[:, :, t]).astype(np.float64)
            sabs = np.abs(amk) @ np.abs(B[:, :, t].astype(np.float64))
            o64 = O[:, :, t].astype(np.float64)
            tol = 2.0 * U.f32_gate(K, sabs) + 2.0 ** -11 * np.abs(o64) + 2.0 ** -25
            err = np.abs(wo.view(got, so)[:, :, t].astype(np.float64) - o64)
            assert (err <= tol).all(), f"f16 gemm {M}x{K}x{N} mat {t} tr={tr} vs the f32 restatement on f16 operands: worst err/tol {(err / tol).max():.3g}"


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("M,K,N,alpha", [(8192, 256, 8192, 1.0), (8192, 320, 8192, 1.0), (4096, 1024, 8192, -0.375), (5120, 640, 5120, 1.0), (6144, 448, 6144, 1.0),
                                             (4352, 1024, 4096, 3.0), (4096, 4160, 4352, 1.0), (16384, 256, 4352, 1.0)])
def test_gemm_f16_continuous_walk_is_bit_identical(gpu, M, K, N, alpha, tr):
    _continuous_walk_case(gpu, M, K, N, 1, alpha, tr)


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("M,K,N", [(8200, 512, 8192), (8192, 512, 8193), (4352 + 248, 256, 4096 + 129), (5000, 1024, 5001), (4104, 320, 4488), (16384 + 8, 256, 4097)])
def test_gemm_f16_continuous_walk_ragged_tiles(gpu, M, K, N, tr):
    """... and with a ragged last tile row and / or column (M % 8 == 0 is the fast path's condition; N is free): the DMA offsets of an edge tile clamp the rows past the
    end to the last valid one, from the slot where the operand's cursor enters the tile; its epilogue skips them; a wave of an edge tile may store nothing at all (N % 256
    <= 128), which is why the stores of a ragged tile are waited out instead of being counted. Same bits as the per-tile launch, nothing written past the end."""
    _continuous_walk_case(gpu, M, K, N, 1, 1.0, tr)


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("M,K,N,mats", [(2048, 512, 2048, 5), (1024, 1024, 1280, 17), (4096, 320, 2048, 3), (1000, 512, 1100, 23)])
def test_gemm_f16_continuous_walk_through_a_batch(gpu, M, K, N, mats, tr):
    """... and through the matrices of a batch in turn (the per-tile launch's grid.y, flattened into the walk's ids): 64 x 5, 20 x 17 and 128 x 3 tiles."""
    _continuous_walk_case(gpu, M, K, N, mats, 1.0, tr)


def _walk_fuzz_cases():
    rng = np.random.default_rng(20261003)
    cases = []
    while len(cases) < 18:
        tm, tn, mats = int(rng.integers(3, 40)), int(rng.integers(3, 40)), int(rng.choice([1, 1, 2, 3]))
        if not 256 < tm * tn * mats <= 900:
            continue
        ragged_m, ragged_n = (8 * int(rng.integers(1, 32)) if rng.integers(0, 2) else 0), (int(rng.integers(1, 256)) if rng.integers(0, 2) else 0)
        cases.append((256 * tm - ragged_m, 64 * int(rng.integers(4, 14)), 256 * tn - ragged_n, mats, bool(rng.integers(0, 2))))
    return cases


@pytest.mark.parametrize("M,K,N,mats,tr", _walk_fuzz_cases())
def test_gemm_f16_continuous_walk_fuzz(gpu, M, K, N, mats, tr):
    """Seeded random tile grids (more than one round of tiles, up to 3.5; half of them with a ragged last tile row and / or column), 4 .. 13 stages per tile, one to
    three matrices, both variants: the walk's bits are the
    per-tile launch's (a tile boundary falls on a different stage of the DMA ring, of A's half-stage slots and of the cut-up tail's plan in nearly every case)."""
    _continuous_walk_case(gpu, M, K, N, mats, 1.0, tr)


def _continuous_walk_case(gpu, M, K, N, mats, alpha, tr):
    """f16 Gemm / GemmTr on the continuous tile walk (gemm_f16.hip m16_cont: one workgroup per CU goes from tile to tile without stopping its LDS-DMA stream; the default
    for K <= 4096 -- K <= 8192 below 16 rounds -- on more than one round of whole tiles) computes every tile exactly as the per-tile launch does: same bits -- whole rounds, a ragged last round, a cut-up
    tail behind the full rounds (6144^2: 64 tiles left over; 4352 x 4096: 16), 4 and 5 stages per tile (the shortest the walk takes), alpha != 1 (an f16-denormal
    result included: the f32 product is rounded once) -- and every element written exactly once (NaN pre-fill); the default rule must give those bits too."""
    wg = _wg()
    rng = np.random.default_rng(M + K + N)
    a = (rng.random(M * K * mats, dtype=np.float32) * 2 - 1).astype(np.float16)
    b = (rng.random(K * N * mats, dtype=np.float32) * 2 - 1).astype(np.float16)
    m1 = upload(gpu, (K, M, mats) if tr else (M, K, mats), a, np.float16)
    m2 = upload(gpu, (K, N, mats), b, np.float16)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    res = {}
    old = {k: gpu.get_tuning(k) for k in ("f16_cont", "f16_tile")}
    try:
        gpu.set_tuning("f16_tile", 256)  # (the 256 x 128 pairs would take the shortest K otherwise)
        for cont in (0, 1, -1):
            gpu.set_tuning("f16_cont", cont)
            out = upload(gpu, (M, N, mats), np.full(M * N * mats, np.nan, np.float16), np.float16)
            run_pass(gpu, lambda p: gemm.dispatch_ex(gpu.device(), shapes, p, alpha, 0.0, out, m1, m2, variant))
            res[cont] = out.read(gpu.device()).view(np.uint16).copy()
    finally:
        for k, v in old.items():
            gpu.set_tuning(k, v)
    assert not np.isnan(res[0].view(np.float16)).any()
    assert np.array_equal(res[1], res[0]), f"continuous walk differs from the per-tile launch in {(res[1] != res[0]).sum()} elements"
    assert np.array_equal(res[-1], res[0])
    # a sample of the output against f64 on the same f16 operands (the per-tile kernel's own parity tests cover it in full at smaller sizes)
    z = mats - 1  # (the last matrix of the batch)
    az, bz = a.reshape(mats, -1)[z], b.reshape(mats, -1)[z]
    B = bz.reshape(N, K).astype(np.float64)   # column-major (K, N): column n of B is row n here
    rows, cols = rng.integers(0, M, 64), rng.integers(0, N, 64)
    Arows = az.reshape(M, K)[rows] if tr else az.reshape(K, M)[:, rows].T  # row m of op(A)
    exact = alpha * np.einsum("ik,ik->i", Arows.astype(np.float64), B[cols])
    got = res[1].view(np.float16).reshape(mats, N, M)[z][cols, rows].astype(np.float64)
    bound = np.abs(exact) * 2.0 ** -11 + abs(alpha) * K * 2.0 ** -22 + 2.0 ** -24
    assert (np.abs(got - exact) <= bound).all()


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("M,K,N", [(1280, 1024, 768), (2056, 448, 1000), (256, 192, 4096)])
def test_gemm_f16_tile_scheduler_is_bit_identical(gpu, M, K, N, tr):
    """The cross-XCD tile scheduler (workgroups take their tile from per-XCD queues; on by default from 16 rounds of tiles) changes
    WHERE a tile runs, never what it computes: forced on a few-tile shape (ragged edges included) it must reproduce the static
    launch bit for bit -- and every tile must have been written exactly once (NaN pre-fill)."""
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(M + K + N + int(tr))
    a = (rng.random(M * K, dtype=np.float32) * 2 - 1).astype(np.float16)
    b = (rng.random(K * N, dtype=np.float32) * 2 - 1).astype(np.float16)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    m1 = upload(gpu, (K, M, 1) if tr else (M, K, 1), a, np.float16)
    m2 = upload(gpu, (K, N, 1), b, np.float16)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    res = {}
    old = {k: gpu.get_tuning(k) for k in ("f16_sched", "f16_tile")}
    try:
        gpu.set_tuning("f16_tile", 256)
        for sched in ("0", "1", "1"):
            gpu.set_tuning("f16_sched", int(sched))
            out = upload(gpu, (M, N, 1), np.full(M * N, np.nan, np.float16), np.float16)
            run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, m1, m2, variant))
            res.setdefault(sched, []).append(out.read(gpu.device()).view(np.uint16).copy())
    finally:
        for k, v in old.items():
            gpu.set_tuning(k, v)
    assert not np.isnan(res["0"][0].view(np.float16)).any()
    for r in res["1"]:
        assert np.array_equal(r, res["0"][0])
    A = wo.view(a, wo.Shape(K, M, 1) if tr else wo.Shape(M, K, 1))[:, :, 0]
    f16_check(res["1"][0].view(np.float16).reshape(N, M).T, (A.T if tr else A).astype(np.float64), wo.view(b, wo.Shape(K, N, 1))[:, :, 0].astype(np.float64), K, "scheduler")


def test_gemm_f16_balance_units_see_fresh_accumulators_every_launch(gpu):
    """The prefix -> suffix hand-off reuses the same scratch tiles and flags launch after launch, written on one XCD and read on another
    whose L2 may still hold last launch's lines: back-to-back launches on DIFFERENT operands (small enough for everything to stay
    cached) must each reproduce their own static result bit for bit -- stale accumulators or a stale flag would show here."""
    wg = _wg()
    M = N = 4096  # 256 tiles: one workgroup per CU, no split-K; everything stays in the caches between launches
    K = 512
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    rng = np.random.default_rng(99)
    sets = []
    for i in range(4):
        a = (rng.random(M * K, dtype=np.float32) * 2 - 1).astype(np.float16)
        b = (rng.random(K * N, dtype=np.float32) * 2 - 1).astype(np.float16)
        sets.append((upload(gpu, (M, K, 1), a, np.float16), upload(gpu, (K, N, 1), b, np.float16)))
    old = {k: gpu.get_tuning(k) for k in ("f16_sched", "f16_tile", "f16_balance")}
    try:
        gpu.set_tuning("f16_tile", 256)
        gpu.set_tuning("f16_sched", 0)
        ref, got = [], []
        gpu.set_tuning("f16_balance", 0)
        for m1, m2 in sets:
            out = upload(gpu, (M, N, 1), np.full(M * N, np.nan, np.float16), np.float16)
            run_pass(gpu, lambda p: gemm.dispatch(gpu.device(), shapes, p, out, m1, m2))
            ref.append(out.read(gpu.device()).view(np.uint16).copy())
        gpu.set_tuning("f16_balance", 1)
        n_before = gpu.f16_balance_info()["balanced_launches"]
        outs = [upload(gpu, (M, N, 1), np.full(M * N, np.nan, np.float16), np.float16) for _ in range(3 * len(sets))]

        def burst(p):  # one pass, no synchronisation between the launches: operands change from launch to launch
            for j, out in enumerate(outs):
                m1, m2 = sets[j % len(sets)]
                gemm.dispatch(gpu.device(), shapes, p, out, m1, m2)
        run_pass(gpu, burst)
        got = [o.read(gpu.device()).view(np.uint16).copy() for o in outs]
        assert gpu.f16_balance_info()["balanced_launches"] == n_before + len(outs)  # the launches really ran as prefix / suffix units
    finally:
        for k, v in old.items():
            gpu.set_tuning(k, v)
    for j, g_ in enumerate(got):
        assert np.array_equal(g_, ref[j % len(sets)]), f"launch {j}: {(g_ != ref[j % len(sets)]).sum()} elements differ from the static launch"


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("M,K,N", [(4096, 1024, 4096), (4104, 768, 4352), (8192, 512, 2048), (2048, 2048, 8448)])  # >= 256 tiles: no split-K
def test_gemm_f16_balance_units_are_bit_identical(gpu, M, K, N, tr):
    """Calibrated shares across XCDs: a fast XCD's workgroups run the first stages of K of a slow XCD's tiles (raw f32 accumulators to
    scratch + a flag) and the owner continues from those accumulators -- the same k-ordered chain in the same registers, so the result
    must equal the unsplit launch bit for bit. WG_TUNE_F16_BALANCE = 1 forces a fixed pattern of takers and givers (two giving rounds
    included) on shapes of a few tiles per XCD, ragged edges included; every element must have been written (NaN pre-fill), and repeated
    launches (flag epochs) must agree."""
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(M + K + N + int(tr) + 5)
    a = (rng.random(M * K, dtype=np.float32) * 2 - 1).astype(np.float16)
    b = (rng.random(K * N, dtype=np.float32) * 2 - 1).astype(np.float16)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    m1 = upload(gpu, (K, M, 1) if tr else (M, K, 1), a, np.float16)
    m2 = upload(gpu, (K, N, 1), b, np.float16)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    res = {}
    old = {k: gpu.get_tuning(k) for k in ("f16_sched", "f16_tile", "f16_balance")}
    try:
        gpu.set_tuning("f16_tile", 256)
        gpu.set_tuning("f16_sched", 0)
        n_before = gpu.f16_balance_info()["balanced_launches"]
        for bal in (0, 1, 1, 1):
            gpu.set_tuning("f16_balance", bal)
            out = upload(gpu, (M, N, 1), np.full(M * N, np.nan, np.float16), np.float16)
            run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, m1, m2, variant))
            res.setdefault(bal, []).append(out.read(gpu.device()).view(np.uint16).copy())
        assert gpu.f16_balance_info()["balanced_launches"] == n_before + 3  # the forced launches really ran as prefix / suffix units
    finally:
        for k, v in old.items():
            gpu.set_tuning(k, v)
    assert not np.isnan(res[0][0].view(np.float16)).any()
    for r in res[1]:
        assert np.array_equal(r, res[0][0])
    A = wo.view(a, wo.Shape(K, M, 1) if tr else wo.Shape(M, K, 1))[:, :, 0]
    f16_check(res[1][0].view(np.float16).reshape(N, M).T, (A.T if tr else A).astype(np.float64), wo.view(b, wo.Shape(K, N, 1))[:, :, 0].astype(np.float64), K, "balance")


@pytest.mark.parametrize("tr", [False, True])
def test_gemm_f16_identity_asymmetric(gpu, f16_tile, tr):
    """A = I, asymmetric small-integer B (exact in f16): any row/column permutation or transposition in the tr-read,
    the interleaved tile map or the epilogue shows up as a bit mismatch."""
    wg = _wg()
    M = K = 512
    N = 256
    eye = np.eye(M, K, dtype=np.float16)
    B = ((np.arange(K)[:, None] * 7 + np.arange(N)[None, :] * 13) % 2039 - 1000).astype(np.float16)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    m1, m2 = upload(gpu, (M, K), eye, np.float16), upload(gpu, (K, N), B, np.float16)
    out = upload(gpu, (M, N), np.zeros(M * N, np.float16), np.float16)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, m1, m2, variant))
    got = out.read(gpu.device()).reshape(M, N, order="F")
    assert np.array_equal(got, B), f"mismatch at {np.argwhere(got != B)[:5]}"
    # and a permutation matrix on the left (row p(i) of B lands in row i): catches M-side mapping errors the identity hides
    perm = np.random.default_rng(3).permutation(M)
    P = np.zeros((M, K), np.float16)
    P[np.arange(M), perm] = 1
    m1p = upload(gpu, (M, K), P.T if tr else P, np.float16)
    run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, m1p, m2, variant))
    got = out.read(gpu.device()).reshape(M, N, order="F")
    assert np.array_equal(got, B[perm]), f"mismatch at {np.argwhere(got != B[perm])[:5]}"


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_axpy(gpu, oracle_c, dtype):
    """Extension: y = fma(alpha, x, y); alpha = +-1 must give the bits of OpAssign Add / Sub."""
    wg, wo = _wg(), _wo()
    n, off_y, off_x = 100003, 3, 8
    rng = np.random.default_rng(11)
    py = (rng.random(n + 16, dtype=np.float32) * 4 - 2).astype(dtype)
    px = (rng.random(n + 16, dtype=np.float32) * 4 - 2).astype(dtype)
    shapes, axpy = wg.ViewShapeBuffers(), wg.Axpy.from_device(gpu.device())
    for alpha in (1.0, -1.0, 0.0, 0.3, -2.5e3):
        ty, tx = upload(gpu, (n + 16,), py, dtype), upload(gpu, (n + 16,), px, dtype)
        run_pass(gpu, lambda p: axpy.dispatch(gpu.device(), shapes, p, alpha, ty.rows(off_y, n), tx.rows(off_x, n)))
        got = ty.read(gpu.device())
        exp32 = py.astype(np.float32)
        oracle_c.axpy(alpha, exp32, wo.Shape(n, 1, 1, 1, 1, off_y), px.astype(np.float32), wo.Shape(n, 1, 1, 1, 1, off_x))
        with np.errstate(over="ignore"):
            exp = exp32.astype(dtype)
        U.assert_bits_equal(got, exp, f"axpy alpha={alpha} {np.dtype(dtype).name}")
        if alpha in (1.0, -1.0):
            t2 = upload(gpu, (n + 16,), py, dtype)
            op = wg.OpAssign.new(gpu.device(), wg.OpAssignVariant.Add if alpha > 0 else wg.OpAssignVariant.Sub)
            run_pass(gpu, lambda p: op.dispatch(gpu.device(), shapes, p, t2.rows(off_y, n), tx.rows(off_x, n)))
            U.assert_bits_equal(got, t2.read(gpu.device()), f"axpy(alpha={alpha}) vs OpAssign")
    with pytest.raises(wg.DimensionMismatch, match="Axpy: dimension mismatch."):
        enc = gpu.device().create_command_encoder()
        axpy.dispatch(gpu.device(), shapes, enc.compute_pass("e", None), 1.0, ty.rows(0, 8), tx.rows(0, 12))


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("M,K,N,mats", [(256, 64, 256, 1), (264, 96, 136, 2), (512, 4096, 256, 1), (36, 20, 28, 1)])
def test_gemm_ex_alpha_beta(gpu, dtype, M, K, N, mats):
    """Extension: out = alpha * A B + beta * out; (1, 0) is bit-identical to Gemm::dispatch; beta = 0 ignores NaNs in out."""
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(M + K + N)
    a = (rng.random(M * K * mats, dtype=np.float32) * 2 - 1).astype(dtype)
    b = (rng.random(K * N * mats, dtype=np.float32) * 2 - 1).astype(dtype)
    c0 = (rng.random(M * N * mats, dtype=np.float32) * 2 - 1).astype(dtype)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    m1, m2 = upload(gpu, (M, K, mats), a, dtype), upload(gpu, (K, N, mats), b, dtype)
    plain = upload(gpu, (M, N, mats), np.full(M * N * mats, np.nan, dtype), dtype)
    run_pass(gpu, lambda p: gemm.dispatch(gpu.device(), shapes, p, plain, m1, m2))
    ref_plain = plain.read(gpu.device())
    ex10 = upload(gpu, (M, N, mats), np.full(M * N * mats, np.nan, dtype), dtype)  # beta = 0: NaNs must not leak
    run_pass(gpu, lambda p: gemm.dispatch_ex(gpu.device(), shapes, p, 1.0, 0.0, ex10, m1, m2))
    U.assert_bits_equal(ex10.read(gpu.device()), ref_plain, "gemm_ex(1, 0) vs gemm")
    for alpha, beta in ((0.5, 0.0), (1.0, 1.0), (-1.5, 0.25)):
        out = upload(gpu, (M, N, mats), c0, dtype)
        run_pass(gpu, lambda p: gemm.dispatch_ex(gpu.device(), shapes, p, alpha, beta, out, m1, m2))
        got = wo.view(out.read(gpu.device()), wo.Shape(M, N, mats)).astype(np.float64)
        A, B, C0 = wo.view(a, wo.Shape(M, K, mats)), wo.view(b, wo.Shape(K, N, mats)), wo.view(c0, wo.Shape(M, N, mats))
        for t in range(mats):
            a64, b64 = A[:, :, t].astype(np.float64), B[:, :, t].astype(np.float64)
            truth = alpha * (a64 @ b64) + beta * C0[:, :, t].astype(np.float64)
            sabs = abs(alpha) * (np.abs(a64) @ np.abs(b64)) + abs(beta) * np.abs(C0[:, :, t]).astype(np.float64)
            tol = U.f32_gate(K + 2, sabs) + (2.0 ** -11 * np.abs(truth) + 2.0 ** -25 if dtype == np.float16 else 0)
            err = np.abs(got[:, :, t] - truth)
            assert (err <= tol).all(), f"gemm_ex({alpha},{beta}) {np.dtype(dtype).name}: worst err/tol {(err / tol).max():.3g}"


@pytest.mark.parametrize("tr", [False, True])
def test_gemv_strided_views(gpu, oracle_c, tr):
    """m = rows/columns sub-view of a bigger matrix in a cube, v / out = offset views with a column stride > length, 3 RHS."""
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(55 + tr)
    PR, PC = 200, 160
    pm = (rng.random(PR * PC * 2, dtype=np.float32) - 0.5).astype(np.float32)
    tm = upload(gpu, (PR, PC, 2), pm)
    m_view = tm.as_view().matrix(1).columns(8, 96).rows(12, 128)  # 128 x 96 inside matrix 1
    R, C = 128, 96
    vlen, olen = (R, C) if tr else (C, R)
    pv = (rng.random(4 + 3 * (vlen + 8), dtype=np.float32) - 0.5).astype(np.float32)
    po = rng.random(8 + 3 * (olen + 4), dtype=np.float32)
    tv, to = upload(gpu, (pv.size,), pv), upload(gpu, (po.size,), po)
    v_view = wg.GpuTensorView(wg.ViewShape((vlen, 3, 1), vlen + 8, 1, 4), tv, 2)
    o_view = wg.GpuTensorView(wg.ViewShape((olen, 3, 1), olen + 4, 1, 8), to, 2)
    variant = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv
    gemv, shapes = wg.Gemv.from_device(gpu.device()), wg.ViewShapeBuffers()
    run_pass(gpu, lambda p: gemv.dispatch_generic(gpu.device(), shapes, p, o_view, m_view, v_view, variant))
    got = to.read(gpu.device())
    sh = lambda v: wo.Shape(v.shape().size[0], v.shape().size[1], v.shape().size[2], v.shape().stride, v.shape().stride_mat, v.shape().offset)
    orc = po.copy()
    oracle_c.gemv(int(variant), orc, sh(o_view), pm, sh(m_view), pv, sh(v_view))
    A, X = wo.view(pm, sh(m_view))[:, :, 0], wo.view(pv, sh(v_view))[:, :, 0]
    A = A.T if tr else A
    truth, sabs = wo.gemm_f64(A, X)
    U.assert_close_f64(wo.view(got, sh(o_view))[:, :, 0], truth, vlen, sabs, "strided gemv vs f64")
    U.assert_close_oracle(wo.view(got, sh(o_view))[:, :, 0], wo.view(orc, sh(o_view))[:, :, 0], vlen, sabs, "strided gemv vs oracle")
    mask = np.ones(po.size, bool)
    s = sh(o_view).resolved()
    mask[(s.offset + np.arange(olen)[:, None] + np.arange(3)[None, :] * s.stride).ravel()] = False
    assert np.array_equal(got[mask], po[mask]), "gemv wrote outside its output view"


def test_reduce_long_vectors_bit_exact(gpu, oracle_c):
    """Few long vectors take the 8-wave LDS-ring kernel: ragged lengths, several vectors, a strided matrix view -- all bit-exact."""
    wg, wo = _wg(), _wo()
    shapes = wg.ViewShapeBuffers()
    rng = np.random.default_rng(123)
    n, nvec, ld = 300_007, 5, 300_012  # ragged tail (n % 128 = 103), rows left over after whole 32-row slots, padded columns
    x = (rng.random(ld * nvec + 8, dtype=np.float32) * 2 - 1).astype(np.float32)
    tx = upload(gpu, (x.size,), x)
    view = wg.GpuTensorView(wg.ViewShape((n, nvec, 1), ld, ld * nvec, 4), tx, 2)
    for op in wg.ReduceOp:
        red = wg.Reduce.new(gpu.device(), op)
        res = upload(gpu, (nvec,), np.full(nvec, np.nan, np.float32))
        run_pass(gpu, lambda p: red.dispatch_batched(gpu.device(), shapes, p, view, res))
        exp = oracle_c.reduce_batched(int(op), x, wo.Shape(n, nvec, 1, ld, ld * nvec, 4))
        U.assert_bits_equal(res.read(gpu.device()), exp, f"long-vector reduce {op!r}")
        one = upload(gpu, (), np.array([np.nan], np.float32))
        v1 = wg.GpuTensorView(wg.ViewShape((1 << 20, 1, 1), 1, 1, 8), tx, 1)
        run_pass(gpu, lambda p: red.dispatch(gpu.device(), shapes, p, v1, one))
        U.assert_bits_equal(one.read(gpu.device()), np.array([oracle_c.reduce(int(op), x, wo.Shape(1 << 20, 1, 1, 1, 1, 8))], np.float32),
                            f"single 2^20 vector {op!r}")


# --------------------------------------------------------------------------------------------------------
# ROW_MAJOR operator surface (shape.wgsl:49-57, linalg/shape.rs:11-15)
# --------------------------------------------------------------------------------------------------------
def _rm_view(wg, t, rows, cols, mats=1, stride=None, stride_mat=None, offset=0):
    """A ROW-major view: index = t*stride_mat + offset + i*stride + j."""
    stride = cols if stride is None else stride
    stride_mat = rows * stride if stride_mat is None else stride_mat
    return wg.GpuTensorView(wg.ViewShape([rows, cols, mats], stride, stride_mat, offset), t, 3)


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("M,K,N,mats", [(64, 128, 32, 1), (260, 72, 132, 2), (512, 256, 768, 1)])
def test_gemm_row_major(gpu, dtype, tr, M, K, N, mats):
    """out = m1 m2 and out = m1^T m2 on ROW-major operands (numpy's default layout), against f64."""
    wg = _wg()
    rng = np.random.default_rng(M + K + N + mats + tr)
    a = (rng.random((mats, K, M) if tr else (mats, M, K), dtype=np.float32) * 2 - 1).astype(dtype)
    b = (rng.random((mats, K, N), dtype=np.float32) * 2 - 1).astype(dtype)
    ta, tb = upload(gpu, (a.size,), a.reshape(-1), dtype), upload(gpu, (b.size,), b.reshape(-1), dtype)
    tc = upload(gpu, (mats * M * N,), np.full(mats * M * N, np.nan, dtype), dtype)
    gemm = wg.Gemm.from_device(gpu.device(), wg.row_major_shader_defs())
    shapes = wg.ViewShapeBuffers()
    va = _rm_view(wg, ta, K, M, mats) if tr else _rm_view(wg, ta, M, K, mats)
    vb, vc = _rm_view(wg, tb, K, N, mats), _rm_view(wg, tc, M, N, mats)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, vc, va, vb, variant))
    got = tc.read(gpu.device()).reshape(mats, M, N).astype(np.float64)
    for t in range(mats):
        a64 = (a[t].T if tr else a[t]).astype(np.float64)
        b64 = b[t].astype(np.float64)
        truth, sabs = a64 @ b64, np.abs(a64) @ np.abs(b64)
        tol = U.f32_gate(K, sabs) + (2.0 ** -11 * np.abs(truth) + 2.0 ** -25 if dtype == np.float16 else 0)
        err = np.abs(got[t] - truth)
        assert (err <= tol).all(), f"row-major gemm tr={tr} mat {t}: worst err/tol {(err / tol).max():.3g}"


@pytest.mark.parametrize("M,K,N,mats,pad", [(4096, 1024, 4096, 1, 0), (4100, 264, 4104, 1, 4), (2048, 528, 2304, 2, 8), (260, 72, 132, 3, 0)])
def test_gemm_tr_row_major_f32_native_kernel_matches_the_transposed_copy(gpu, M, K, N, mats, pad):
    """The same for f32 (gemm_f32.hip's B_NC tile bodies: m2 of the column-major product contiguous along N): bit-identical to the transposed copy + the 256 x 128
    tile kernel wherever that path runs the same unsplit tiles (forced here: no mid-size family, no few-column kernel), and inside the f64 gate everywhere.
    Ragged tiles (the register-staged edge path), a K that is not a multiple of 16 (the remainder tile), strided views with an offset, batches."""
    wg = _wg()
    rng = np.random.default_rng(2 * M + 3 * K + 5 * N + mats)
    sa, sb, sc = M + pad, N + pad, N + 2 * pad
    off = 4 * pad
    a = rng.random((mats, K, sa), dtype=np.float32) * 2 - 1
    b = rng.random((mats, K, sb), dtype=np.float32) * 2 - 1
    ta = upload(gpu, (off + a.size,), np.concatenate([np.zeros(off, np.float32), a.reshape(-1)]))
    tb = upload(gpu, (off + b.size,), np.concatenate([np.zeros(off, np.float32), b.reshape(-1)]))
    gemm = wg.Gemm.from_device(gpu.device(), wg.row_major_shader_defs())
    shapes = wg.ViewShapeBuffers()
    va = _rm_view(wg, ta, K, M, mats, stride=sa, stride_mat=K * sa, offset=off)
    vb = _rm_view(wg, tb, K, N, mats, stride=sb, stride_mat=K * sb, offset=off)
    res = {}
    olds = {k: gpu.set_tuning(k, 0) for k in ("f32_mid", "f32_skinny", "f32_panels")}  # the copy path on the big tile too: the same accumulation chains
    try:
        for native in (1, 0):
            tc = upload(gpu, (off + mats * M * sc,), np.full(off + mats * M * sc, np.nan, np.float32))
            vc = _rm_view(wg, tc, M, N, mats, stride=sc, stride_mat=M * sc, offset=off)
            old = gpu.set_tuning("rm_tr_native", native)
            try:
                run_pass(gpu, lambda p: gemm.dispatch_tr(gpu.device(), shapes, p, vc, va, vb))
            finally:
                gpu.set_tuning("rm_tr_native", old)
            res[native] = tc.read(gpu.device())
    finally:
        for k, v in olds.items():
            gpu.set_tuning(k, v)
    full = res[1][off:].reshape(mats, M, sc)
    assert np.isnan(full[:, :, N:]).all() and np.isnan(res[1][:off]).all(), "elements outside the output view were written"
    got = full[:, :, :N].astype(np.float64)
    for t in range(mats):
        a64, b64 = a[t, :, :M].T.astype(np.float64), b[t, :, :N].astype(np.float64)
        truth, sabs = a64 @ b64, np.abs(a64) @ np.abs(b64)
        err = np.abs(got[t] - truth)
        assert (err <= U.f32_gate(K, sabs)).all(), f"row-major f32 GemmTr mat {t}: worst err/tol {(err / U.f32_gate(K, sabs)).max():.3g}"
    tiles = (M + 255) // 256 * ((N + 127) // 128) * mats  # (output^T is N x M: 256 x 128 tiles of it; the count is symmetric enough for this purpose)
    if ((N + 255) // 256 * ((M + 127) // 128) * mats) % 256 == 0 and tiles >= 256:  # whole rounds: the copy path's launcher cuts no K either (no tail split)
        U.assert_bits_equal(res[1], res[0], "row-major f32 GemmTr: native kernel vs transposed copy")


@pytest.mark.parametrize("tile", [0, 128, 256128, 256])  # the launcher's own choice, then every tile family forced (gemm_f16_t128.hip's B_NC instances / gemm_f16_nt.hip)
@pytest.mark.parametrize("M,K,N,mats,pad", [(256, 256, 256, 1, 0), (264, 320, 520, 2, 8), (1024, 1024, 768, 1, 0), (2048, 512, 4096, 1, 16), (8, 256, 8, 3, 0),
                                            (2304, 2048, 1280, 1, 0)])
def test_gemm_tr_row_major_f16_native_kernel_matches_the_transposed_copy(gpu, M, K, N, mats, pad, tile):
    """Row-major GemmTr of f16 operands (shape.wgsl:49-57, gemm.wgsl:115-148): the kernel that takes m1 where it lies (gemm_f16_nt.hip: both operands
    contiguous along their output dimension, no scratch) against the transposed-copy path of round 5 -- the same k order and accumulation chains, so the
    bits must agree -- and against f64. Strided row-major views with an offset, batches, ragged tiles (M, N not multiples of 256), the shortest K."""
    wg = _wg()
    rng = np.random.default_rng(M + 3 * K + 5 * N + mats)
    sa, sb, sc = M + pad, N + pad, N + 2 * pad                     # row strides of m1 (K x M), m2 (K x N), out (M x N)
    off = 8 * pad                                                  # (multiples of 8 elements: 16-byte alignment is the kernels' fast-path condition)
    a = (rng.random((mats, K, sa), dtype=np.float32) * 2 - 1).astype(np.float16)
    b = (rng.random((mats, K, sb), dtype=np.float32) * 2 - 1).astype(np.float16)
    ta = upload(gpu, (off + a.size,), np.concatenate([np.zeros(off, np.float16), a.reshape(-1)]), np.float16)
    tb = upload(gpu, (off + b.size,), np.concatenate([np.zeros(off, np.float16), b.reshape(-1)]), np.float16)
    gemm = wg.Gemm.from_device(gpu.device(), wg.row_major_shader_defs())
    shapes = wg.ViewShapeBuffers()
    va = _rm_view(wg, ta, K, M, mats, stride=sa, stride_mat=K * sa, offset=off)
    vb = _rm_view(wg, tb, K, N, mats, stride=sb, stride_mat=K * sb, offset=off)
    res = {}
    for native in (1, 0):
        tc = upload(gpu, (off + mats * M * sc,), np.full(off + mats * M * sc, np.nan, np.float16), np.float16)
        vc = _rm_view(wg, tc, M, N, mats, stride=sc, stride_mat=M * sc, offset=off)
        old, old_tile = gpu.set_tuning("rm_tr_native", native), gpu.set_tuning("f16_tile", tile)
        try:
            run_pass(gpu, lambda p: gemm.dispatch_tr(gpu.device(), shapes, p, vc, va, vb))
        finally:
            gpu.set_tuning("rm_tr_native", old)
            gpu.set_tuning("f16_tile", old_tile)
        res[native] = tc.read(gpu.device())
    # (the 256 x 256 family forced on an output of fewer tiles than CUs: the copy path's launcher cuts K over the idle CUs there -- other partial sums, other bits; the
    # native kernel never cuts K. Everywhere else the two ways run the same accumulation chains.)
    if not (tile == 256 and -(-M // 256) * -(-N // 256) * mats < 256):
        U.assert_bits_equal(res[1], res[0], "row-major GemmTr: native kernel vs transposed copy")
    full = res[1][off:].reshape(mats, M, sc)
    assert np.isnan(full[:, :, N:]).all() and np.isnan(res[1][:off]).all(), "elements outside the output view were written"
    got = full[:, :, :N].astype(np.float64)
    for t in range(mats):
        a64, b64 = a[t, :, :M].T.astype(np.float64), b[t, :, :N].astype(np.float64)
        truth, sabs = a64 @ b64, np.abs(a64) @ np.abs(b64)
        tol = U.f32_gate(K, sabs) + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25
        err = np.abs(got[t] - truth)
        assert (err <= tol).all(), f"row-major GemmTr mat {t}: worst err/tol {(err / tol).max():.3g}"


@pytest.mark.parametrize("seed", range(8))
def test_gemm_tr_row_major_fuzz(gpu, seed):
    """Random shapes through wg_gemm_rm(GemmTr), f16 and f32, whatever path the launcher takes (the 256 x 256 / mid-size n-contiguous-B kernels, the f32 B_NC tile bodies,
    the transposed copy for what they do not take: K % 64 != 0, small outputs), against f64. Ragged M / N, batches, K from one stage to a few."""
    wg = _wg()
    rng = np.random.default_rng(1000 + seed)
    for dtype in (np.float16, np.float32):
        q = 8 if dtype == np.float16 else 4
        M, N = int(rng.integers(1, 300)) * q, int(rng.integers(1, 300)) * q
        K = int(rng.choice([64, 128, 256, 320, 576, 1024, 1096, 2048])) if dtype == np.float16 else int(rng.integers(1, 200)) * 4
        mats = int(rng.choice([1, 1, 2, 3]))
        a = (rng.random((mats, K, M), dtype=np.float32) * 2 - 1).astype(dtype)
        b = (rng.random((mats, K, N), dtype=np.float32) * 2 - 1).astype(dtype)
        ta, tb = upload(gpu, (a.size,), a.reshape(-1), dtype), upload(gpu, (b.size,), b.reshape(-1), dtype)
        tc = upload(gpu, (mats * M * N,), np.full(mats * M * N, np.nan, dtype), dtype)
        gemm = wg.Gemm.from_device(gpu.device(), wg.row_major_shader_defs())
        shapes = wg.ViewShapeBuffers()
        run_pass(gpu, lambda p: gemm.dispatch_tr(gpu.device(), shapes, p, _rm_view(wg, tc, M, N, mats), _rm_view(wg, ta, K, M, mats), _rm_view(wg, tb, K, N, mats)))
        got = tc.read(gpu.device()).reshape(mats, M, N).astype(np.float64)
        for t in range(mats):
            a64, b64 = a[t].T.astype(np.float64), b[t].astype(np.float64)
            truth, sabs = a64 @ b64, np.abs(a64) @ np.abs(b64)
            tol = U.f32_gate(K, sabs) + (2.0 ** -11 * np.abs(truth) + 2.0 ** -25 if dtype == np.float16 else 0)
            err = np.abs(got[t] - truth)
            assert (err <= tol).all(), f"row-major GemmTr fuzz seed {seed} {np.dtype(dtype).name} {M}x{N}x{K}x{mats} mat {t}: worst err/tol {(err / tol).max():.3g}"


def test_gemm_row_major_strided_view_and_errors(gpu):
    """A row-major sub-view (row stride > cols, offset) of a larger buffer, and the reference's dimension panic."""
    wg = _wg()
    rng = np.random.default_rng(11)
    big = (rng.random((96, 80), dtype=np.float32) * 2 - 1)
    tbig = upload(gpu, (big.size,), big.reshape(-1), np.float32)
    a = big[8:8 + 64, 12:12 + 32]                      # 64 x 32 block, row stride 80, offset 8*80 + 12
    b = (rng.random((32, 48), dtype=np.float32) * 2 - 1)
    tb = upload(gpu, (b.size,), b.reshape(-1), np.float32)
    tc = upload(gpu, (64 * 48,), np.zeros(64 * 48, np.float32), np.float32)
    gemm = wg.Gemm.from_device(gpu.device(), wg.row_major_shader_defs())
    shapes = wg.ViewShapeBuffers()
    va = _rm_view(wg, tbig, 64, 32, 1, stride=80, offset=8 * 80 + 12)
    run_pass(gpu, lambda p: gemm.dispatch(gpu.device(), shapes, p, _rm_view(wg, tc, 64, 48), va, _rm_view(wg, tb, 32, 48)))
    got = tc.read(gpu.device()).reshape(64, 48).astype(np.float64)
    truth, sabs = a.astype(np.float64) @ b.astype(np.float64), np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert (np.abs(got - truth) <= U.f32_gate(32, sabs)).all()
    with pytest.raises(wg.DimensionMismatch, match="Gemm: dimension mismatch."):
        run_pass(gpu, lambda p: gemm.dispatch(gpu.device(), shapes, p, _rm_view(wg, tc, 64, 48), va, _rm_view(wg, tb, 48, 32)))


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("R,Cn", [(64, 256), (1024, 1024), (260, 72)])
def test_gemv_row_major(gpu, tr, R, Cn):
    wg = _wg()
    rng = np.random.default_rng(R + Cn + tr)
    m = rng.random((R, Cn), dtype=np.float32) * 2 - 1
    vlen, olen = (R, Cn) if tr else (Cn, R)
    v = rng.random(vlen, dtype=np.float32) * 2 - 1
    tm, tv = upload(gpu, (m.size,), m.reshape(-1), np.float32), upload(gpu, (vlen,), v, np.float32)
    to = upload(gpu, (olen,), np.full(olen, np.nan, np.float32), np.float32)
    gemv = wg.Gemv.from_device(gpu.device(), wg.row_major_shader_defs())
    shapes = wg.ViewShapeBuffers()
    variant = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv
    run_pass(gpu, lambda p: gemv.dispatch_generic(gpu.device(), shapes, p, to, _rm_view(wg, tm, R, Cn), tv, variant))
    got = to.read(gpu.device()).astype(np.float64)
    m64 = (m.T if tr else m).astype(np.float64)
    truth, sabs = m64 @ v.astype(np.float64), np.abs(m64) @ np.abs(v).astype(np.float64)
    assert (np.abs(got - truth) <= U.f32_gate(vlen, sabs)).all()


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("R,Cn,nrhs,mats", [(64, 256, 4, 1), (1024, 512, 8, 1), (260, 72, 12, 2), (2048, 640, 32, 1)])
def test_gemv_row_major_multi_rhs(gpu, tr, R, Cn, nrhs, mats):
    """Row-major Gemv with several right-hand-side columns (shape.wgsl:49-57 `im` x grid.y of gemv.wgsl:40,46,62): out[:, y, z] =
    op(m[:, :, z]) v[:, y, z] with every view row-major. Checked against f64 per (matrix, column)."""
    wg = _wg()
    rng = np.random.default_rng(R + Cn + nrhs + tr)
    m = rng.random((mats, R, Cn), dtype=np.float32) * 2 - 1          # [z][i][j]: row-major matrices, stride = Cn
    vlen, olen = (R, Cn) if tr else (Cn, R)
    v = rng.random((mats, vlen, nrhs), dtype=np.float32) * 2 - 1     # row-major (vlen x nrhs): stride = nrhs
    tm, tv = upload(gpu, (m.size,), m.reshape(-1), np.float32), upload(gpu, (v.size,), v.reshape(-1), np.float32)
    to = upload(gpu, (mats * olen * nrhs,), np.full(mats * olen * nrhs, np.nan, np.float32), np.float32)
    gemv = wg.Gemv.from_device(gpu.device(), wg.row_major_shader_defs())
    shapes = wg.ViewShapeBuffers()
    variant = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv
    mv = wg.GpuTensorView(wg.ViewShape([R, Cn, mats], Cn, R * Cn, 0), tm, 3)
    vv = wg.GpuTensorView(wg.ViewShape([vlen, nrhs, mats], nrhs, vlen * nrhs, 0), tv, 3)
    ov = wg.GpuTensorView(wg.ViewShape([olen, nrhs, mats], nrhs, olen * nrhs, 0), to, 3)
    run_pass(gpu, lambda p: gemv.dispatch_generic(gpu.device(), shapes, p, ov, mv, vv, variant))
    got = to.read(gpu.device()).astype(np.float64).reshape(mats, olen, nrhs)
    for z in range(mats):
        m64 = (m[z].T if tr else m[z]).astype(np.float64)
        truth, sabs = m64 @ v[z].astype(np.float64), np.abs(m64) @ np.abs(v[z]).astype(np.float64)
        assert (np.abs(got[z] - truth) <= U.f32_gate(vlen, sabs)).all(), f"row-major Gemv x{nrhs}: matrix {z}"


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("M,K,N", [(512, 192 + 8, 512), (512, 192 + 32, 264), (776, 256 + 40, 520), (512, 448 + 56, 512), (4352, 1024 + 24, 4352), (256, 8192 + 16, 256),
                                   (4096, 512 + 48, 4096)])
def test_gemm_f16_k_remainder(gpu, M, K, N, tr):
    """K % 64 != 0 (any multiple of 8) on the 16x16x32 kernel: the remainder is the loop's stage 0 -- zero-padded through the LDS by
    ordinary loads + stores in the DMA's image -- on plain launches, ragged tiles, the tail split (17 x 17 tiles), split-K (the last
    split takes it) and, forced, the balance units (the prefix unit multiplies it). Poisoned neighbours: the k-values just past K in
    memory (the next column / row block) are NaN, so a remainder stage that reads one element too many cannot pass."""
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(M + K + N + int(tr))
    KP = K + 8  # parents with 8 extra k, filled with NaN
    if tr:   # m1 stored K x M: k runs down the columns -> pad rows
        pa = np.full((M, KP), np.nan, np.float16); pa[:, :K] = (rng.random((M, K), dtype=np.float32) * 2 - 1).astype(np.float16)
        a_flat, a_view_shape = pa.reshape(-1), wg.ViewShape((K, M, 1), KP, KP * M, 0)
        A64 = pa[:, :K].astype(np.float64)
    else:    # m1 stored M x K column-major: k indexes columns -> extra NaN columns behind
        pa = np.full((KP, M), np.nan, np.float16); pa[:K] = (rng.random((K, M), dtype=np.float32) * 2 - 1).astype(np.float16)
        a_flat, a_view_shape = pa.reshape(-1), wg.ViewShape((M, K, 1), M, M * KP, 0)
        A64 = pa[:K].T.astype(np.float64)
    pb = np.full((N, KP), np.nan, np.float16); pb[:, :K] = (rng.random((N, K), dtype=np.float32) * 2 - 1).astype(np.float16)
    B64 = pb[:, :K].T.astype(np.float64)
    ta, tb = upload(gpu, (a_flat.size,), a_flat, np.float16), upload(gpu, (pb.size,), pb.reshape(-1), np.float16)
    va = wg.GpuTensorView(a_view_shape, ta, 2)
    vb = wg.GpuTensorView(wg.ViewShape((K, N, 1), KP, KP * N, 0), tb, 2)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    res = []
    old = {k: gpu.get_tuning(k) for k in ("f16_tile", "f16_balance", "f16_sched")}
    try:
        for tile, bal, sched in ((0, 0, -1), (256, 0, 0), (256, 1, 0), (256, 0, 1)):
            gpu.set_tuning("f16_tile", tile); gpu.set_tuning("f16_balance", bal); gpu.set_tuning("f16_sched", sched)
            out = upload(gpu, (M, N, 1), np.full(M * N, np.nan, np.float16), np.float16)
            run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, va, vb, variant))
            res.append(out.read(gpu.device()).reshape(N, M).T.copy())
    finally:
        for k, v in old.items():
            gpu.set_tuning(k, v)
    for r in res:
        f16_check(r, A64, B64, K, f"K remainder {M}x{K}x{N} tr={tr}")
    # static map, balance units and the tile scheduler run the same accumulation chains: bit-identical
    assert np.array_equal(res[1].view(np.uint16), res[2].view(np.uint16)) and np.array_equal(res[1].view(np.uint16), res[3].view(np.uint16))


@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("M,K,N", [(128, 72, 128), (128, 136, 256), (1000, 1000, 1000), (520, 328, 264), (256, 4104, 128), (2048, 2056, 2048), (136, 120, 72)])
def test_gemm_f16_k_remainder_small_tiles(gpu, M, K, N, tr):
    """The same on the 128 x 128 kernel (outputs of fewer than one 256 x 256 tile per CU, K from 72 on): the remainder is its half-stages 0 and 1;
    under split-K the last split owns it. Forced (wg_ctx_set_tuning) and on the launcher's own choice; NaN just past K in memory."""
    wg = _wg()
    rng = np.random.default_rng(M * 3 + K + N + int(tr))
    KP = K + 8
    if tr:
        pa = np.full((M, KP), np.nan, np.float16); pa[:, :K] = (rng.random((M, K), dtype=np.float32) * 2 - 1).astype(np.float16)
        a_flat, a_view_shape = pa.reshape(-1), wg.ViewShape((K, M, 1), KP, KP * M, 0)
        A64 = pa[:, :K].astype(np.float64)
    else:
        pa = np.full((KP, M), np.nan, np.float16); pa[:K] = (rng.random((K, M), dtype=np.float32) * 2 - 1).astype(np.float16)
        a_flat, a_view_shape = pa.reshape(-1), wg.ViewShape((M, K, 1), M, M * KP, 0)
        A64 = pa[:K].T.astype(np.float64)
    pb = np.full((N, KP), np.nan, np.float16); pb[:, :K] = (rng.random((N, K), dtype=np.float32) * 2 - 1).astype(np.float16)
    B64 = pb[:, :K].T.astype(np.float64)
    ta, tb = upload(gpu, (a_flat.size,), a_flat, np.float16), upload(gpu, (pb.size,), pb.reshape(-1), np.float16)
    va = wg.GpuTensorView(a_view_shape, ta, 2)
    vb = wg.GpuTensorView(wg.ViewShape((K, N, 1), KP, KP * N, 0), tb, 2)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    old = gpu.get_tuning("f16_tile")
    try:
        for tile in (128, 256128, 0):
            gpu.set_tuning("f16_tile", tile)
            out = upload(gpu, (M, N, 1), np.full(M * N, np.nan, np.float16), np.float16)
            run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, va, vb, variant))
            f16_check(out.read(gpu.device()).reshape(N, M).T.copy(), A64, B64, K, f"K remainder, 128-tiles {tile}: {M}x{K}x{N} tr={tr}")
    finally:
        gpu.set_tuning("f16_tile", old)


# --------------------------------------------------------------------------------------------------------
# seeded fuzz over the f16 MFMA paths: random ragged sizes, strides, offsets, batches, both variants, alpha/beta
# --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(24))
def test_gemm_f16_fuzz(gpu, f16_tile, seed):
    wg = _wg()
    rng = np.random.default_rng(1000 + seed)
    M, N = int(rng.integers(1, 80)) * 8, int(rng.integers(1, 80)) * 8
    K = int(rng.choice([192, 256, 320, 448, 576, 1024, 96, 160, 2048 + 64, 200, 232, 248, 440, 1000, 64, 128, 72]))  # whole stages, K % 64 remainders of every size, short K
    mats = int(rng.choice([1, 1, 2, 3]))
    tr = bool(rng.integers(0, 2))
    ex = bool(rng.integers(0, 2))
    alpha, beta = (float(rng.choice([1.0, -1.0, 0.5, 2.0])), float(rng.choice([0.0, 1.0, -0.5]))) if ex else (1.0, 0.0)
    # operand views inside larger buffers: leading dimension padded by a multiple of 8, offset a multiple of 8 (16-byte aligned)
    def view_of(rows, cols):
        ld = rows + 8 * int(rng.integers(0, 3))
        sm = ld * cols + 8 * int(rng.integers(0, 3))
        off = 8 * int(rng.integers(0, 4))
        data = (rng.random(off + sm * mats, dtype=np.float32) * 2 - 1).astype(np.float16)
        return data, ld, sm, off
    ar, ac = (K, M) if tr else (M, K)
    a, lda, sma, offa = view_of(ar, ac)
    b, ldb, smb, offb = view_of(K, N)
    c, ldc, smc, offc = view_of(M, N)
    ta, tb, tc = upload(gpu, (a.size,), a, np.float16), upload(gpu, (b.size,), b, np.float16), upload(gpu, (c.size,), c, np.float16)
    mk = lambda t, r, cc, ld, sm, off: wg.GpuTensorView(wg.ViewShape([r, cc, mats], ld, sm, off), t, 3)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    va, vb, vc = mk(ta, ar, ac, lda, sma, offa), mk(tb, K, N, ldb, smb, offb), mk(tc, M, N, ldc, smc, offc)
    if ex:
        run_pass(gpu, lambda p: gemm.dispatch_ex(gpu.device(), shapes, p, alpha, beta, vc, va, vb, variant))
    else:
        run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, vc, va, vb, variant))
    got = tc.read(gpu.device())

    def mat(buf, rows, cols, ld, sm, off, t):
        idx = off + t * sm + np.arange(rows)[:, None] + np.arange(cols)[None, :] * ld
        return buf[idx]
    touched = np.zeros(c.size, bool)
    for t in range(mats):
        A = mat(a, ar, ac, lda, sma, offa, t).astype(np.float64)
        A = A.T if tr else A
        B = mat(b, K, N, ldb, smb, offb, t).astype(np.float64)
        C0 = mat(c, M, N, ldc, smc, offc, t).astype(np.float64)
        truth = alpha * (A @ B) + beta * C0
        sabs = abs(alpha) * (np.abs(A) @ np.abs(B)) + abs(beta) * np.abs(C0)
        tol = U.f32_gate(K + 2, sabs) + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25
        idx = offc + t * smc + np.arange(M)[:, None] + np.arange(N)[None, :] * ldc
        err = np.abs(got[idx].astype(np.float64) - truth)
        assert (err <= tol).all(), f"fuzz {seed}: M={M} N={N} K={K} mats={mats} tr={tr} ab=({alpha},{beta}) worst {(err / tol).max():.3g}"
        touched[idx.reshape(-1)] = True
    # nothing outside the output view was written (padding between columns / matrices, the offset prefix)
    assert np.array_equal(got[~touched], c[~touched]), f"fuzz {seed}: bytes outside the output view changed"


@pytest.mark.parametrize("seed", range(16))
def test_gemm_f16_fuzz_odd_shapes(gpu, seed):
    """Sizes, leading dimensions and offsets that are multiples of 4 (the operator's vec4 precondition, shape.wgsl:64-66) but not of 8: the zero-padded staging path in front of the MFMA kernels, with (alpha, beta) and batches; checks the values and
    that nothing outside the output view is written."""
    wg = _wg()
    rng = np.random.default_rng(3000 + seed)
    M, N, K = 4 * int(rng.integers(33, 175)), 4 * int(rng.integers(33, 175)), 4 * int(rng.integers(65, 375))
    mats = int(rng.choice([1, 1, 2]))
    tr = bool(rng.integers(0, 2))
    ex = bool(rng.integers(0, 2))
    alpha, beta = (float(rng.choice([1.0, -1.0, 0.5])), float(rng.choice([0.0, 1.0, -0.5]))) if ex else (1.0, 0.0)

    def view_of(rows, cols):
        ld = rows + 4 * int(rng.integers(0, 3))
        sm = ld * cols + 4 * int(rng.integers(0, 3))
        off = 4 * int(rng.integers(0, 4))
        data = (rng.random(off + sm * mats, dtype=np.float32) * 2 - 1).astype(np.float16)
        return data, ld, sm, off
    ar, ac = (K, M) if tr else (M, K)
    a, lda, sma, offa = view_of(ar, ac)
    b, ldb, smb, offb = view_of(K, N)
    c, ldc, smc, offc = view_of(M, N)
    ta, tb, tc = upload(gpu, (a.size,), a, np.float16), upload(gpu, (b.size,), b, np.float16), upload(gpu, (c.size,), c, np.float16)
    mk = lambda t, r, cc, ld, sm, off: wg.GpuTensorView(wg.ViewShape([r, cc, mats], ld, sm, off), t, 3)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    va, vb, vc = mk(ta, ar, ac, lda, sma, offa), mk(tb, K, N, ldb, smb, offb), mk(tc, M, N, ldc, smc, offc)
    if ex:
        run_pass(gpu, lambda p: gemm.dispatch_ex(gpu.device(), shapes, p, alpha, beta, vc, va, vb, variant))
    else:
        run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, vc, va, vb, variant))
    got = tc.read(gpu.device())
    touched = np.zeros(c.size, bool)
    for t in range(mats):
        mat = lambda buf, rows, cols, ld, sm, off: buf[off + t * sm + np.arange(rows)[:, None] + np.arange(cols)[None, :] * ld]
        A = mat(a, ar, ac, lda, sma, offa).astype(np.float64)
        A = A.T if tr else A
        B = mat(b, K, N, ldb, smb, offb).astype(np.float64)
        C0 = mat(c, M, N, ldc, smc, offc).astype(np.float64)
        truth = alpha * (A @ B) + beta * C0
        sabs = abs(alpha) * (np.abs(A) @ np.abs(B)) + abs(beta) * np.abs(C0)
        tol = U.f32_gate(K + 2, sabs) + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25
        idx = offc + t * smc + np.arange(M)[:, None] + np.arange(N)[None, :] * ldc
        err = np.abs(got[idx].astype(np.float64) - truth)
        assert (err <= tol).all(), f"odd fuzz {seed}: M={M} N={N} K={K} mats={mats} tr={tr} ab=({alpha},{beta}) worst {(err / tol).max():.3g}"
        touched[idx.reshape(-1)] = True
    assert np.array_equal(got[~touched], c[~touched]), f"odd fuzz {seed}: bytes outside the output view changed"


@pytest.mark.parametrize("seed", range(12))
def test_gemm_f32_fuzz(gpu, seed):
    """Same idea for the f32 kernel (vec4 granularity: everything a multiple of 4), incl. split-K-sized K and (alpha, beta)."""
    rng = np.random.default_rng(2000 + seed)
    M, N = int(rng.integers(1, 130)) * 4, int(rng.integers(1, 100)) * 4
    K = int(rng.choice([4, 16, 64, 132, 512, 1024, 4096 + 16]))
    _gemm_f32_fuzz_case(gpu, rng, seed, M, N, K, int(rng.choice([1, 1, 2])), bool(rng.integers(0, 2)))


@pytest.mark.parametrize("seed", range(10))
def test_gemm_f32_fuzz_few_columns(gpu, seed):
    """N <= 64 with many rows: the streaming kernel of gemm_f32_skinny.hip -- with and without a K split (direct output with alpha / beta
    when the row blocks alone fill the chip), ragged row blocks, K % 8 == 4, padded views, batches; GemmTr of the same shapes stays on
    the tiled kernel."""
    rng = np.random.default_rng(4000 + seed)
    M = 4 * int(rng.integers(128, 600)) if seed % 2 else 4 * int(rng.integers(8200, 9000))  # 33-36 k rows: no K split
    N = 4 * int(rng.integers(1, 17))
    K = int(rng.choice([128, 132, 260, 1000, 2052]))
    _gemm_f32_fuzz_case(gpu, rng, seed, M, N, K, int(rng.choice([1, 1, 2])), bool(rng.integers(0, 4) == 0))


@pytest.mark.parametrize("seed", range(8))
def test_gemm_f32_fuzz_column_panels(gpu, seed):
    """64-column panels of the few-column kernel (what small squares run on), forced for ragged sizes, batches, both variants, alpha / beta."""
    rng = np.random.default_rng(6000 + seed)
    M, N = 4 * int(rng.integers(32, 300)), 4 * int(rng.integers(17, 200))
    K = int(rng.choice([128, 132, 516, 1024]))
    old = gpu.set_tuning("f32_panels", 1)
    try:
        _gemm_f32_fuzz_case(gpu, rng, seed, M, N, K, int(rng.choice([1, 1, 2])), bool(rng.integers(0, 2)))
    finally:
        gpu.set_tuning("f32_panels", old)


@pytest.mark.parametrize("seed", range(10))
def test_gemm_f32_fuzz_few_rows(gpu, seed):
    """M <= 64 with many columns: computed transposed on the few-column GemmTr kernel (operands read in place, output written transposed by
    the epilogue or the strided split-K reduce) -- both variants, padded views, batches, alpha; beta != 0 falls back to the tiles."""
    rng = np.random.default_rng(5000 + seed)
    M = 4 * int(rng.integers(1, 17))
    N = 4 * int(rng.integers(128, 400)) if seed % 2 else 4 * int(rng.integers(8200, 8600))  # ~33 k columns: no K split
    K = int(rng.choice([128, 132, 260, 1000, 2052]))
    _gemm_f32_fuzz_case(gpu, rng, seed, M, N, K, int(rng.choice([1, 1, 2])), bool(rng.integers(0, 2)))


def _gemm_f32_fuzz_case(gpu, rng, seed, M, N, K, mats, tr):
    wg = _wg()
    ex = bool(rng.integers(0, 2))
    alpha, beta = (float(rng.choice([1.0, -1.0, 0.5])), float(rng.choice([0.0, 1.0, -0.5]))) if ex else (1.0, 0.0)
    def view_of(rows, cols):
        ld = rows + 4 * int(rng.integers(0, 3))
        sm = ld * cols + 4 * int(rng.integers(0, 3))
        off = 4 * int(rng.integers(0, 4))
        return (rng.random(off + sm * mats, dtype=np.float32) * 2 - 1), ld, sm, off
    ar, ac = (K, M) if tr else (M, K)
    a, lda, sma, offa = view_of(ar, ac)
    b, ldb, smb, offb = view_of(K, N)
    c, ldc, smc, offc = view_of(M, N)
    ta, tb, tc = upload(gpu, (a.size,), a), upload(gpu, (b.size,), b), upload(gpu, (c.size,), c)
    mk = lambda t, r, cc, ld, sm, off: wg.GpuTensorView(wg.ViewShape([r, cc, mats], ld, sm, off), t, 3)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    va, vb, vc = mk(ta, ar, ac, lda, sma, offa), mk(tb, K, N, ldb, smb, offb), mk(tc, M, N, ldc, smc, offc)
    if ex:
        run_pass(gpu, lambda p: gemm.dispatch_ex(gpu.device(), shapes, p, alpha, beta, vc, va, vb, variant))
    else:
        run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, vc, va, vb, variant))
    got = tc.read(gpu.device())
    touched = np.zeros(c.size, bool)
    for t in range(mats):
        ia = offa + t * sma + np.arange(ar)[:, None] + np.arange(ac)[None, :] * lda
        A = a[ia].astype(np.float64)
        A = A.T if tr else A
        B = b[offb + t * smb + np.arange(K)[:, None] + np.arange(N)[None, :] * ldb].astype(np.float64)
        idx = offc + t * smc + np.arange(M)[:, None] + np.arange(N)[None, :] * ldc
        C0 = c[idx].astype(np.float64)
        truth = alpha * (A @ B) + beta * C0
        sabs = abs(alpha) * (np.abs(A) @ np.abs(B)) + abs(beta) * np.abs(C0)
        err = np.abs(got[idx].astype(np.float64) - truth)
        tol = U.f32_gate(K + 2, sabs)
        assert (err <= tol).all(), f"f32 fuzz {seed}: M={M} N={N} K={K} mats={mats} tr={tr} ab=({alpha},{beta}) worst {(err / tol).max():.3g}"
        touched[idx.reshape(-1)] = True
    assert np.array_equal(got[~touched], c[~touched]), f"f32 fuzz {seed}: bytes outside the output view changed"


def test_record_replay_gemm_chain(gpu):
    """A recorded command buffer (hipGraph) holding a dependent chain of GEMMs, incl. a split-K f16 GEMM (workspace), a row-major
    GemmTr (transpose scratch) and a GEMV: replaying it with new inputs gives the same result as eager dispatch."""
    wg = _wg()
    dev, shapes = gpu.device(), wg.ViewShapeBuffers()
    rng = np.random.default_rng(77)
    M, K, N = 256, 4096, 256  # few tiles, long K: split-K path
    a = (rng.random(M * K, dtype=np.float32) * 2 - 1).astype(np.float16) / 16
    b = (rng.random(K * N, dtype=np.float32) * 2 - 1).astype(np.float16) / 16
    ta, tb = upload(gpu, (M, K), a, np.float16), upload(gpu, (K, N), b, np.float16)
    t1 = upload(gpu, (M, N), np.zeros(M * N, np.float16), np.float16)      # t1 = a b
    t2 = upload(gpu, (M, N), np.zeros(M * N, np.float16), np.float16)      # t2 = t1^T t1 (row-major GemmTr)
    tv = upload(gpu, (N,), (rng.random(N, dtype=np.float32)).astype(np.float32))
    t1f = upload(gpu, (M, N), np.zeros(M * N, np.float32))                 # f32 copy target for the GEMV
    to = upload(gpu, (M,), np.zeros(M, np.float32))
    gemm = wg.Gemm.from_device(dev)
    gemm_rm = wg.Gemm.from_device(dev, wg.row_major_shader_defs())
    gemv = wg.Gemv.from_device(dev)
    rm = lambda t, r, c: wg.GpuTensorView(wg.ViewShape([r, c, 1], c, r * c, 0), t, 3)

    def chain(p):
        gemm.dispatch(dev, shapes, p, t1, ta, tb)
        gemm_rm.dispatch_tr(dev, shapes, p, rm(t2, M, N), rm(t1, M, N), rm(t1, M, N))  # the same memory read as row-major M x N
        gemv.dispatch(dev, shapes, p, to, t1f, tv)

    run_pass(gpu, chain)  # eager once: sizes the workspaces, gives the expected values
    exp1, exp2, exp_o = t1.read(dev).copy(), t2.read(dev).copy(), to.read(dev).copy()
    enc = dev.create_command_encoder(record=True)
    with enc.compute_pass("recorded", None) as p:
        chain(p)
    cb = enc.finish()
    zero = wg.OpAssign.new(dev, wg.OpAssignVariant.Sub)  # clear the outputs on the device (x -= x) so that the replay has to recompute them
    run_pass(gpu, lambda p: (zero.dispatch(dev, shapes, p, to, to), None)[1])
    assert not to.read(dev).any()
    for _ in range(3):
        gpu.queue().submit([cb])
    assert np.array_equal(t1.read(dev), exp1) and np.array_equal(t2.read(dev), exp2) and np.array_equal(to.read(dev), exp_o)
    # sanity of the chain itself against f64
    A, B = a.reshape(M, K, order="F").astype(np.float64), b.reshape(K, N, order="F").astype(np.float64)
    T1 = exp1.reshape(M, N, order="F").astype(np.float64)
    assert np.abs(T1 - A @ B).max() <= 2.0 ** -9 * np.abs(A @ B).max() + 1e-3
    T1rm = exp1.reshape(M, N).astype(np.float64)  # the same bytes as a row-major M x N matrix
    T2 = exp2.reshape(M, N).astype(np.float64)
    ref = T1rm.T @ T1rm
    assert np.abs(T2 - ref).max() <= 2.0 ** -9 * np.abs(ref).max() + 1e-3


def test_scratch_regrow_keeps_recorded_command_buffers_valid():
    """A recorded command buffer has the context's scratch pointer baked into its graph and may be submitted many times. (1) An
    operator that would have to GROW the scratch inside a recording returns the distinct WG_ERR_WORKSPACE status (allocation cannot
    be captured) and the recording stays usable. (2) Growing the scratch AFTER a recording must not free the region the old command
    buffer replays into: the region is retired, so a buffer allocated afterwards can never alias it (canary), and the replay is right."""
    wg = _wg()
    inst = wg.GpuInstance.new(0)  # a fresh context: empty scratch
    dev, shapes = inst.device(), wg.ViewShapeBuffers()
    R, Cn = 4096, 16384  # split over grid.y: partial sums go through the scratch
    rng = np.random.default_rng(5)
    m, v = rng.random(R * Cn, dtype=np.float32), rng.random(Cn, dtype=np.float32)
    S = wg.BufferUsages
    mk = lambda shape, flat: wg.TensorBuilder.tensor(shape, S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, flat)
    tm, tv, to = mk((R, Cn), m), mk((Cn,), v), mk((R,), np.zeros(R, np.float32))
    gemv = wg.Gemv.from_device(dev)
    enc = dev.create_command_encoder(record=True)
    with enc.compute_pass("too early", None) as p:
        with pytest.raises(wg.WorkspaceMustGrow) as ei:
            gemv.dispatch(dev, shapes, p, to, tm, tv)
        assert ei.value.status == 8 and "while recording" in str(ei.value)
    enc.finish()  # the recording itself is still well-formed (it holds nothing)
    p = dev.create_command_encoder().compute_pass("eager", None)
    gemv.dispatch(dev, shapes, p, to, tm, tv)  # eager once: sizes the scratch
    expect = to.read(dev).copy()
    enc = dev.create_command_encoder(record=True)
    with enc.compute_pass("recorded", None) as p2:
        gemv.dispatch(dev, shapes, p2, to, tm, tv)
    cb = enc.finish()
    dev.reserve_workspace(96 << 20)  # a later, larger need: the scratch regrows while `cb` is alive
    canaries = [mk(((1 << 20) * k // 4,), np.full((1 << 20) * k // 4, 7.0, np.float32)) for k in (1, 1, 2, 2, 4, 4, 8)]  # would reuse a freed region
    wg.OpAssign.new(dev, wg.OpAssignVariant.Sub).dispatch(dev, shapes, p, to, to)
    assert not to.read(dev).any()
    for _ in range(3):
        inst.queue().submit([cb])
    assert np.array_equal(to.read(dev), expect)
    for c in canaries:
        assert (c.read(dev) == 7.0).all(), "a replayed command buffer wrote into memory that was freed and reallocated"
    del cb, canaries
    inst.close()


# --------------------------------------------------------------------------------------------------------
# two-pass multi-workgroup reduce (extension): Min/Max bit-identical, Sum/Prod/SqNorm within the re-association bound
# --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,off", [(0, 0), (1, 0), (5, 3), (4097, 1), (65536, 0), (300007, 2), ((1 << 22) + 3, 5)])
def test_reduce_fast(gpu, oracle_c, n, off):
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(n + off)
    x = np.concatenate([np.zeros(off, np.float32), (rng.random(n, dtype=np.float32) * 2 - 1)]).astype(np.float32)
    if n:
        x[off + rng.integers(0, n)] = -0.0
    tx = upload(gpu, (x.size,), x) if x.size else None
    if tx is None:
        return
    view = wg.GpuTensorView(wg.ViewShape([n, 1, 1], n, n, off), tx, 1)
    x64 = x[off:].astype(np.float64)
    for op, wop in ((wg.ReduceOp.Min, wo.MIN), (wg.ReduceOp.Max, wo.MAX), (wg.ReduceOp.Sum, wo.SUM), (wg.ReduceOp.SqNorm, wo.SQNORM),
                    (wg.ReduceOp.Prod, wo.PROD)):
        res = upload(gpu, (), np.full(1, np.nan, np.float32))
        red = wg.Reduce.new(gpu.device(), op)
        run_pass(gpu, lambda p: red.dispatch_fast(gpu.device(), wg.ViewShapeBuffers(), p, view, res))
        got = res.read(gpu.device())[0]
        ref = np.float32(oracle_c.reduce(wop, x, wo.Shape(n, 1, 1, n, n, off)))
        if op in (wg.ReduceOp.Min, wg.ReduceOp.Max):
            assert got.tobytes() == ref.tobytes(), f"fast {op.name} n={n}: {got} vs {ref}"
        elif op == wg.ReduceOp.Sum:
            assert abs(float(got) - x64.sum()) <= max(n, 1) * 2.0 ** -24 * np.abs(x64).sum() + 1e-30
            assert abs(float(got) - float(ref)) <= 2 * max(n, 1) * 2.0 ** -24 * np.abs(x64).sum() + 1e-30
        elif op == wg.ReduceOp.SqNorm:
            assert abs(float(got) - (x64 * x64).sum()) <= max(n, 1) * 2.0 ** -23 * (x64 * x64).sum() + 1e-30
        else:  # Prod of |x| < 1 underflows to 0 quickly; for short vectors compare relatively
            truth = np.prod(x64)
            assert abs(float(got) - truth) <= max(n, 1) * 2.0 ** -23 * abs(truth) + 1e-37
        # determinism: a second run gives the same bits
        run_pass(gpu, lambda p: red.dispatch_fast(gpu.device(), wg.ViewShapeBuffers(), p, view, res))
        assert res.read(gpu.device())[0].tobytes() == got.tobytes()


@pytest.mark.parametrize("n", [65536, 65540, 300007, (1 << 22) + 8])
@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_reduce_min_max_long_vector_bits(gpu, oracle_c, n, dtype):
    """Min / Max of one long vector run on the two-pass kernels (order-free operations); the same data as TWO columns of a batched call runs
    the reference-order kernels: the bits must agree -- signed zeros as the extreme value included (the min of a non-negative vector holding
    both zeros, the max of a non-positive one) -- and agree with the oracle where the oracle's own answer does not depend on the order."""
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(n)
    base = rng.random(n, dtype=np.float32).astype(dtype)
    for sign, op, wop in ((1.0, wg.ReduceOp.Min, wo.MIN), (-1.0, wg.ReduceOp.Max, wo.MAX), (1.0, wg.ReduceOp.Max, wo.MAX), (-1.0, wg.ReduceOp.Min, wo.MIN)):
        x = (base * dtype(sign)).astype(dtype)
        idx = rng.integers(0, n, 64)
        x[idx[:32]] = dtype(0.0)
        x[idx[32:]] = dtype(-0.0)
        two = np.concatenate([x, x])
        t1, t2 = upload(gpu, (n,), x, dtype), upload(gpu, (n, 2), two, dtype)
        r1, r2 = upload(gpu, (), np.full(1, np.nan, dtype), dtype), upload(gpu, (2,), np.full(2, np.nan, dtype), dtype)
        red = wg.Reduce.new(gpu.device(), op)
        run_pass(gpu, lambda p: (red.dispatch(gpu.device(), wg.ViewShapeBuffers(), p, t1, r1), red.dispatch_batched(gpu.device(), wg.ViewShapeBuffers(), p, t2, r2)))
        a, b = r1.read(gpu.device()), r2.read(gpu.device())
        assert a[0].tobytes() == b[0].tobytes() == b[1].tobytes(), f"{op.name} sign {sign} n={n} {np.dtype(dtype).name}: one vector {a[0]!r} vs batched {b!r}"
        if (sign > 0) == (op == wg.ReduceOp.Max):  # the extreme value is not a zero: the oracle's answer is order-free too
            ref = np.float32(oracle_c.reduce(wop, x.astype(np.float32), wo.Shape(n, 1, 1, n, n, 0)))
            assert np.float32(a[0]) == ref


@pytest.mark.parametrize("tr", [False, True])
def test_gemv_reduce(gpu, tr):
    """reduce(op, op(m) v) in one call == Gemv then Reduce, bit for bit."""
    wg = _wg()
    rng = np.random.default_rng(5 + tr)
    R, Cn = 1024, 2048
    m = rng.random(R * Cn, dtype=np.float32) * 2 - 1
    vlen, olen = (R, Cn) if tr else (Cn, R)
    v = rng.random(vlen, dtype=np.float32) * 2 - 1
    tm, tv = upload(gpu, (R, Cn), m), upload(gpu, (vlen,), v)
    to = upload(gpu, (olen,), np.zeros(olen, np.float32))
    variant = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv
    gemv, shapes = wg.Gemv.from_device(gpu.device()), wg.ViewShapeBuffers()
    for op in (wg.ReduceOp.SqNorm, wg.ReduceOp.Max, wg.ReduceOp.Sum):
        r1, r2 = upload(gpu, (), np.zeros(1, np.float32)), upload(gpu, (), np.zeros(1, np.float32))
        red = wg.Reduce.new(gpu.device(), op)
        def two(p):
            gemv.dispatch_generic(gpu.device(), shapes, p, to, tm, tv, variant)
            red.dispatch(gpu.device(), shapes, p, to, r1)
        run_pass(gpu, two)
        run_pass(gpu, lambda p: wg.gemv_reduce(p, op, r2, tm, tv, variant))
        assert r1.read(gpu.device()).tobytes() == r2.read(gpu.device()).tobytes()


@pytest.mark.parametrize("R,Cn", [(128, 4096), (260, 512), (1024, 1024), (4096, 1024), (132, 8), (8192, 1024)])
def test_gemv_reduce_fused_single_launch(gpu, R, Cn):
    """The launch-bound family runs Gemv + Reduce as ONE kernel (last workgroup folds y in the reference order; gemv.hip): same bits as
    the two dispatches for all five operators, on repeated calls (the arrival counter re-arms itself), from a strided / offset matrix
    view, and when replayed from a recorded command buffer. (8192 x 1024 exceeds 4 Mi elements: the two-launch path, same bits.)"""
    wg = _wg()
    dev, shapes = gpu.device(), wg.ViewShapeBuffers()
    rng = np.random.default_rng(R * 7 + Cn)
    ld = R + 8
    m = (rng.random(ld * Cn + 16, dtype=np.float32) * 2 - 1).astype(np.float32)
    v = (rng.random(Cn, dtype=np.float32) + 0.5).astype(np.float32)
    tm, tv = upload(gpu, (m.size,), m), upload(gpu, (Cn,), v)
    mv = wg.GpuTensorView(wg.ViewShape([R, Cn, 1], ld, ld * Cn, 8), tm, 2)  # offset 8, leading dimension R + 8
    to = upload(gpu, (R,), np.zeros(R, np.float32))
    gemv = wg.Gemv.from_device(dev)
    for op in (wg.ReduceOp.Min, wg.ReduceOp.Max, wg.ReduceOp.Sum, wg.ReduceOp.Prod, wg.ReduceOp.SqNorm):
        r1, r2 = upload(gpu, (), np.zeros(1, np.float32)), upload(gpu, (), np.full(1, np.nan, np.float32))
        red = wg.Reduce.new(dev, op)
        run_pass(gpu, lambda p: (gemv.dispatch(dev, shapes, p, to, mv, tv), red.dispatch(dev, shapes, p, to, r1)))
        for _ in range(3):
            run_pass(gpu, lambda p: wg.gemv_reduce(p, op, r2, mv, tv, wg.GemvVariant.Gemv))
            assert r1.read(dev).tobytes() == r2.read(dev).tobytes(), f"{op.name} {R}x{Cn}"
    # recorded: the fused launch replays (its scratch and counter exist: it ran eagerly above)
    r3 = upload(gpu, (), np.full(1, np.nan, np.float32))
    enc = dev.create_command_encoder(record=True)
    with enc.compute_pass("rec", None) as p:
        wg.gemv_reduce(p, wg.ReduceOp.Sum, r3, mv, tv, wg.GemvVariant.Gemv)
    cb = enc.finish()
    rs = upload(gpu, (), np.zeros(1, np.float32))
    run_pass(gpu, lambda p: (gemv.dispatch(dev, shapes, p, to, mv, tv), wg.Reduce.new(dev, wg.ReduceOp.Sum).dispatch(dev, shapes, p, to, rs)))
    for _ in range(2):
        gpu.queue().submit([cb])
    assert r3.read(dev).tobytes() == rs.read(dev).tobytes()


# --------------------------------------------------------------------------------------------------------
# f16 Gemv / Reduce (extension: `T: Pod` is a parameter of every reference operator, the kernels are f32). Contract = the f16 Gemm's:
# f16 elements, f32 arithmetic, one rounding. Reduce keeps the reference ORDER, so it is bit-exact against the f32 oracle run on the
# (exactly converted) inputs and rounded once.
# --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tr", [False, True])
@pytest.mark.parametrize("R,Cn,nrhs,mats", [(1024, 2048, 1, 1), (256, 512, 3, 2), (4096, 8192, 1, 1), (132, 260, 2, 1),
                                           # one right-hand side: 16-byte loads where aligned (k % 8, batches), 8-byte loads otherwise, splits
                                           (12, 12, 1, 3), (1028, 36, 1, 2), (40004, 68, 1, 1), (300000, 8, 1, 1), (16392, 20, 1, 1), (2056, 4100, 1, 1),
                                           # 3 .. 8 right-hand sides past the launch-bound sizes: the f16 Gemm kernels
                                           (8192, 4096, 3, 1), (4096, 2112, 8, 1), (4104, 6152, 5, 1),
                                           # GemvTr with two right-hand sides from 2048 outputs on: the column kernel's 2-vector form (16-byte and 8-byte loads, ragged rows, a batch)
                                           (4096, 4096, 2, 1), (2056, 4100, 2, 1), (1000, 2048, 2, 2), (12292, 2052, 2, 1)])
def test_gemv_f16(gpu, tr, R, Cn, nrhs, mats):
    wg = _wg()
    rng = np.random.default_rng(R + Cn + nrhs + tr)
    m = (rng.random((mats, Cn, R), dtype=np.float32) * 2 - 1).astype(np.float16)       # [z][col][row]: column-major R x Cn
    vlen, olen = (R, Cn) if tr else (Cn, R)
    v = (rng.random((mats, nrhs, vlen), dtype=np.float32) * 2 - 1).astype(np.float16)  # [z][rhs][i]
    tm, tv = upload(gpu, (R, Cn, mats), m.reshape(-1), np.float16), upload(gpu, (vlen, nrhs, mats), v.reshape(-1), np.float16)
    to = upload(gpu, (olen, nrhs, mats), np.full(mats * nrhs * olen, np.nan, np.float16), np.float16)
    gemv, shapes = wg.Gemv.from_device(gpu.device()), wg.ViewShapeBuffers()
    variant = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv
    run_pass(gpu, lambda p: gemv.dispatch_generic(gpu.device(), shapes, p, to, tm, tv, variant))
    got = to.read(gpu.device()).astype(np.float64).reshape(mats, nrhs, olen)
    for z in range(mats):
        M64 = m[z].astype(np.float64).T  # R x Cn
        M64 = M64.T if tr else M64
        truth, sabs = M64 @ v[z].astype(np.float64).T, np.abs(M64) @ np.abs(v[z].astype(np.float64)).T
        tol = U.f32_gate(vlen, sabs) + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25
        assert (np.abs(got[z].T - truth) <= tol).all(), f"f16 Gemv tr={tr} matrix {z}: worst {(np.abs(got[z].T - truth) / tol).max():.3g}"


@pytest.mark.parametrize("n,off", [(0, 0), (1, 0), (127, 0), (128, 4), (129, 1), (345, 3), (65536, 0), (100003, 8)])
def test_reduce_f16_bit_exact(gpu, oracle_c, n, off):
    wg = _wg()
    from oracle import wgsl_oracle as wo
    rng = np.random.default_rng(n + off)
    x = (rng.random(n + off + 8, dtype=np.float32) * 0.5 + 0.75).astype(np.float16)  # around 1: products stay finite
    tx = upload(gpu, (x.size,), x, np.float16)
    view = wg.GpuTensorView(wg.ViewShape([n, 1, 1], max(n, 1), max(n, 1), off), tx, 1)
    for op, o in ((wg.ReduceOp.Min, wo.MIN), (wg.ReduceOp.Max, wo.MAX), (wg.ReduceOp.Sum, wo.SUM), (wg.ReduceOp.Prod, wo.PROD), (wg.ReduceOp.SqNorm, wo.SQNORM)):
        if op == wg.ReduceOp.Prod and n > 2000:
            continue  # overflows f16 either way
        res = upload(gpu, (), np.zeros(1, np.float16), np.float16)
        run_pass(gpu, lambda p: wg.Reduce.new(gpu.device(), op).dispatch(gpu.device(), wg.ViewShapeBuffers(), p, view, res))
        exp32 = np.float32(oracle_c.reduce(o, x.astype(np.float32), wo.Shape(n, 1, 1, max(n, 1), max(n, 1), off)))
        with np.errstate(over="ignore"):
            exp = np.float16(exp32)
        assert res.read(gpu.device())[0].tobytes() == exp.tobytes(), f"f16 reduce {op.name} n={n} off={off}"
    # batched: one f16 result per column
    if n >= 128:
        cols = 5
        xm = (rng.random(n * cols, dtype=np.float32) - 0.5).astype(np.float16)
        tm = upload(gpu, (n, cols), xm, np.float16)
        rs = upload(gpu, (cols,), np.zeros(cols, np.float16), np.float16)
        run_pass(gpu, lambda p: wg.Reduce.new(gpu.device(), wg.ReduceOp.Sum).dispatch_batched(gpu.device(), wg.ViewShapeBuffers(), p, tm, rs))
        got = rs.read(gpu.device())
        for c in range(cols):
            e = np.float16(np.float32(oracle_c.reduce(wo.SUM, xm.astype(np.float32), wo.Shape(n, 1, 1, n, n, c * n))))
            assert got[c].tobytes() == e.tobytes()


@pytest.mark.parametrize("n,off", [(5, 3), (4097, 1), (65536, 0), (300007, 2)])
def test_reduce_fast_f16(gpu, oracle_c, n, off):
    """Two-pass reduce on f16 vectors: Min/Max have the bits of the reference-order Reduce; Sum/SqNorm are re-associated in f32 and rounded
    once: within the f32 re-association bound + half an f16 ulp of the f64 truth."""
    wg = _wg()
    rng = np.random.default_rng(n + off)
    x = (rng.random(n + off + 8, dtype=np.float32) - 0.5).astype(np.float16)
    tx = upload(gpu, (x.size,), x, np.float16)
    view = wg.GpuTensorView(wg.ViewShape([n, 1, 1], n, n, off), tx, 1)
    xs = x[off:off + n].astype(np.float64)
    for op in (wg.ReduceOp.Min, wg.ReduceOp.Max, wg.ReduceOp.Sum, wg.ReduceOp.SqNorm):
        r_ref, r_fast = upload(gpu, (), np.zeros(1, np.float16), np.float16), upload(gpu, (), np.zeros(1, np.float16), np.float16)
        red = wg.Reduce.new(gpu.device(), op)
        run_pass(gpu, lambda p: (red.dispatch(gpu.device(), wg.ViewShapeBuffers(), p, view, r_ref), red.dispatch_fast(gpu.device(), wg.ViewShapeBuffers(), p, view, r_fast)))
        a, b = r_ref.read(gpu.device())[0], r_fast.read(gpu.device())[0]
        if op in (wg.ReduceOp.Min, wg.ReduceOp.Max):
            assert a.tobytes() == b.tobytes()
        else:
            truth = xs.sum() if op == wg.ReduceOp.Sum else (xs * xs).sum()
            sabs = np.abs(xs).sum() if op == wg.ReduceOp.Sum else (xs * xs).sum()
            tol = n * 2.0 ** -24 * sabs + 2.0 ** -10 * abs(truth) + 2.0 ** -24
            assert abs(float(b) - truth) <= tol and abs(float(a) - truth) <= tol, (op, float(a), float(b), truth)


@pytest.mark.parametrize("R,C", [(4096, 1004), (2048, 2500), (512, 68), (1024, 1000)])
@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_gemv_inf_in_the_last_column_stays_inf(gpu, R, C, dtype):
    """A wave's last < 64 columns are loaded in groups (gemv.hip); slots past the end re-load the last valid column and must contribute an exact
    zero -- not (last column) x 0, which is NaN where that column holds an Inf. The reference kernel (gemv.wgsl:28-65) never touches a column
    past the end: +-Inf in the last column gives +-Inf in that row, and no NaN anywhere (round-3 review)."""
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(R + C)
    m = rng.random(R * C, dtype=np.float32).astype(dtype)
    v = (rng.random(C, dtype=np.float32) + np.float32(0.5)).astype(dtype)
    M2 = m.reshape(R, C, order="F")
    M2[5, C - 1], M2[9, C - 1], M2[R - 1, C - 1] = np.inf, -np.inf, np.inf
    tm, tv = upload(gpu, (R, C), m, dtype), upload(gpu, (C,), v, dtype)
    out = upload(gpu, (R,), np.zeros(R, dtype), dtype)
    gemv, shapes = wg.Gemv.from_device(gpu.device()), wg.ViewShapeBuffers()
    run_pass(gpu, lambda p: gemv.dispatch(gpu.device(), shapes, p, out, tm, tv))
    got = out.read(gpu.device()).astype(np.float64)
    assert not np.isnan(got).any(), f"{int(np.isnan(got).sum())} NaN rows"
    assert got[5] == np.inf and got[9] == -np.inf and got[R - 1] == np.inf
    fin = np.ones(R, bool)
    fin[[5, 9, R - 1]] = False
    assert np.isfinite(got[fin]).all()
    if dtype == np.float32:  # the restated kernel agrees on which rows are infinite
        orc = np.zeros(R, np.float32)
        wo.CLib().gemv(wo.GEMV, orc, wo.Shape(R), m, wo.Shape(R, C), v, wo.Shape(C))
        assert np.array_equal(np.isinf(orc), np.isinf(got)) and not np.isnan(orc).any()


# --------------------------------------------------------------------------------------------------------
# The mid-size f32 tile family (gemm_f32_mid.hip, whole K per workgroup): 128 x 128, 128 x 64, 64 x 128 tiles on 2 x 2 waves -- the same k-ordered
# fmaf chain per element whatever the tile: bit for bit against each other and against gemm_f32.hip's unsplit result -- and 64 x 64, 64 x 32,
# 32 x 64 tiles whose four waves split K, (p0 + p2) + (p1 + p3): bit for bit among themselves. Every tile shape forced (WG_TUNE_F32_MID) over
# ragged M / N, a K remainder, batches and both variants, against the oracle and f64.
# --------------------------------------------------------------------------------------------------------
MID_TILES = [128128, 128064, 64128]
MID_KW_TILES = [64064, 64032, 32064, 96096, 96064, 64096]
MID_SHAPES = [(256, 256, 256, 1), (512, 128, 384, 1), (1024, 512, 1024, 1), (132, 64, 68, 1), (260, 200, 324, 3), (64, 32, 64, 5), (1000, 1000, 1000, 1),
              (4, 36, 4, 2), (2048, 48, 2048, 1), (96, 96, 160, 2)]


@pytest.mark.parametrize("M,K,N,mats", MID_SHAPES)
@pytest.mark.parametrize("tr", [False, True])
def test_gemm_f32_mid_tiles(gpu, oracle_c, M, K, N, mats, tr):
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(M * 7919 + K * 31 + N + mats + int(tr))
    a = (rng.random(M * K * mats, dtype=np.float32) * 2 - 1).astype(np.float32)
    b = (rng.random(K * N * mats, dtype=np.float32) * 2 - 1).astype(np.float32)
    s1 = wo.Shape(K, M, mats) if tr else wo.Shape(M, K, mats)
    s2, so = wo.Shape(K, N, mats), wo.Shape(M, N, mats)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    orc = np.zeros(M * N * mats, np.float32)
    oracle_c.gemm(int(variant), orc, so, a, s1, b, s2)
    m1 = upload(gpu, (K, M, mats) if tr else (M, K, mats), a)
    m2 = upload(gpu, (K, N, mats), b)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    A, B = wo.view(a, s1), wo.view(b, s2)
    results = {}
    for knob in MID_TILES + MID_KW_TILES + [1, 0]:  # every tile; the estimate's own tile; the family off (gemm_f32.hip and friends)
        old = gpu.set_tuning("f32_mid", knob)
        try:
            out = upload(gpu, (M, N, mats), np.full(M * N * mats, np.nan, np.float32))
            run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, m1, m2, variant))
            got = out.read(gpu.device())
        finally:
            gpu.set_tuning("f32_mid", old)
        results[knob] = got
        for t in range(mats):
            amk = A[:, :, t].T if tr else A[:, :, t]
            truth, sabs = wo.gemm_f64(amk, B[:, :, t])
            g_t = wo.view(got, so)[:, :, t]
            U.assert_close_f64(g_t, truth, K, sabs, f"mid tile {knob}: gemm {M}x{K}x{N} mat {t} tr={tr} vs f64")
            U.assert_close_oracle(g_t, wo.view(orc, so)[:, :, t], K, sabs, f"mid tile {knob} vs oracle")
    for knob in MID_TILES[1:]:
        U.assert_bits_equal(results[knob], results[MID_TILES[0]], f"mid tile {knob} vs 128 x 128: {M}x{K}x{N} tr={tr}")
    for knob in MID_KW_TILES[1:]:
        U.assert_bits_equal(results[knob], results[MID_KW_TILES[0]], f"k-split tile {knob} vs 64 x 64: {M}x{K}x{N} tr={tr}")
    assert results[1].tobytes() in (results[MID_TILES[0]].tobytes(), results[MID_KW_TILES[0]].tobytes())  # the estimate's tile is one of the two


@pytest.mark.parametrize("tr", [False, True])
def test_gemm_f32_mid_equals_the_big_tile_unsplit(gpu, tr):
    """4096 x 4096 x 256: 512 of gemm_f32.hip's 256 x 128 tiles (no K cut) against the 128 x 128 family: the same chain per element, bit for bit;
    and alpha / beta through the family's epilogue against the two-step form."""
    wg = _wg()
    M = N = 4096
    K = 256
    rng = np.random.default_rng(5 + int(tr))
    a = (rng.random(M * K, dtype=np.float32) * 2 - 1).astype(np.float32)
    b = (rng.random(K * N, dtype=np.float32) * 2 - 1).astype(np.float32)
    c0 = (rng.random(M * N, dtype=np.float32) * 2 - 1).astype(np.float32)
    m1, m2 = upload(gpu, (K, M) if tr else (M, K), a), upload(gpu, (K, N), b)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    got = {}
    for knob in (0, 128128, 64128):
        old = gpu.set_tuning("f32_mid", knob)
        try:
            out = upload(gpu, (M, N), np.full(M * N, np.nan, np.float32))
            run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, m1, m2, variant))
            got[knob] = out.read(gpu.device())
            if knob:
                from wgmath_amd._lib import check, lib
                from wgmath_amd.wgcore import as_view
                oc = upload(gpu, (M, N), c0)
                ov, av, bv = as_view(oc), as_view(m1), as_view(m2)
                check(lib.wg_gemm_ex(gpu._ctx.handle, int(variant), 0, 0.5, -2.0, ov.buffer()._h, ov.shape().to_c(), av.buffer()._h, av.shape().to_c(),
                                     bv.buffer()._h, bv.shape().to_c()))
                exp = np.float32(0.5) * got[knob]
                exp = np.float32(-2.0) * c0 + exp  # fmaf(beta, c, alpha * acc): -2 c is exact, so one rounding either way
                U.assert_bits_equal(oc.read(gpu.device()), exp.astype(np.float32), f"alpha/beta through the mid family (tile {knob})")
        finally:
            gpu.set_tuning("f32_mid", old)
    U.assert_bits_equal(got[128128], got[0], "128 x 128 family vs the 256 x 128 tile")
    U.assert_bits_equal(got[64128], got[0], "64 x 128 tiles vs the 256 x 128 tile")


@pytest.mark.parametrize("M,K,N,mats,split", [(64, 4096, 4096, 1, 4), (4096, 4096, 64, 1, 4), (128, 1000, 192, 2, 3), (64, 96, 64, 1, 2), (256, 2048, 256, 1, 8)])
@pytest.mark.parametrize("tr", [False, True])
def test_gemm_f32_mid_split_k(gpu, oracle_c, M, K, N, mats, split, tr):
    """Few tiles with a long K: the mid family's k-split tiles with K cut across workgroups (f32 slabs + the ordered reduce), forced through
    WG_TUNE_F32_MID / _SPLIT, against the oracle and f64; and once on the launcher's own choice. (A split whose last part would hold no whole
    k-tile is reduced: 64 x 96 x 64 in two parts of 64 + 32.)"""
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(M + 3 * K + 5 * N + mats + int(tr))
    a = (rng.random(M * K * mats, dtype=np.float32) * 2 - 1).astype(np.float32)
    b = (rng.random(K * N * mats, dtype=np.float32) * 2 - 1).astype(np.float32)
    s1 = wo.Shape(K, M, mats) if tr else wo.Shape(M, K, mats)
    s2, so = wo.Shape(K, N, mats), wo.Shape(M, N, mats)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    orc = np.zeros(M * N * mats, np.float32)
    oracle_c.gemm(int(variant), orc, so, a, s1, b, s2)
    m1, m2 = upload(gpu, (K, M, mats) if tr else (M, K, mats), a), upload(gpu, (K, N, mats), b)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    A, B = wo.view(a, s1), wo.view(b, s2)
    for knob, sp in ((64064, split), (64032, split), (-1, 0)):
        old, old_s = gpu.set_tuning("f32_mid", knob), gpu.set_tuning("f32_mid_split", sp)
        try:
            out = upload(gpu, (M, N, mats), np.full(M * N * mats, np.nan, np.float32))
            run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, m1, m2, variant))
            got = out.read(gpu.device())
        finally:
            gpu.set_tuning("f32_mid", old)
            gpu.set_tuning("f32_mid_split", old_s)
        for t in range(mats):
            amk = A[:, :, t].T if tr else A[:, :, t]
            truth, sabs = wo.gemm_f64(amk, B[:, :, t])
            g_t = wo.view(got, so)[:, :, t]
            U.assert_close_f64(g_t, truth, K, sabs, f"mid tile {knob} split {sp}: gemm {M}x{K}x{N} mat {t} tr={tr} vs f64")
            U.assert_close_oracle(g_t, wo.view(orc, so)[:, :, t], K, sabs, f"mid tile {knob} split {sp} vs oracle")


@pytest.mark.parametrize("knob", MID_TILES + MID_KW_TILES)
def test_gemm_f32_mid_strided_views(gpu, oracle_c, knob):
    """Strided / offset views (GpuMatrix::columns / rows / GpuCubeView::matrix) through every tile of the mid family, nothing outside the output
    view touched; and a larger strided case with a K remainder."""
    old = gpu.set_tuning("f32_mid", knob)
    try:
        test_gemm_strided_views(gpu, oracle_c)
        wg, wo = _wg(), _wo()
        rng = np.random.default_rng(knob)
        PR, PC = 400, 360
        pa, pb = (rng.random(PR * PC * 2, dtype=np.float32) - 0.5).astype(np.float32), (rng.random(PR * PC * 2, dtype=np.float32) - 0.5).astype(np.float32)
        po0 = rng.random(PR * PC * 2, dtype=np.float32)
        ta, tb, to = upload(gpu, (PR, PC, 2), pa), upload(gpu, (PR, PC, 2), pb), upload(gpu, (PR, PC, 2), po0)
        M, K, N = 196, 108, 132  # K % 32 = 12, K % 16 = 12: the remainder tile of both sub-families
        a_view = ta.as_view().matrix(1).columns(8, K).rows(12, M)
        b_view = tb.as_view().matrix(0).columns(20, N).rows(4, K)
        o_view = to.as_view().matrix(1).columns(16, N).rows(24, M)
        gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
        run_pass(gpu, lambda p: gemm.dispatch(gpu.device(), shapes, p, o_view, a_view, b_view))
        got = to.read(gpu.device())
        sh = lambda v: wo.Shape(v.shape().size[0], v.shape().size[1], v.shape().size[2], v.shape().stride, v.shape().stride_mat, v.shape().offset)
        A, B = wo.view(pa, sh(a_view))[:, :, 0], wo.view(pb, sh(b_view))[:, :, 0]
        truth, sabs = wo.gemm_f64(A, B)
        U.assert_close_f64(wo.view(got, sh(o_view))[:, :, 0], truth, K, sabs, f"strided gemm, mid tile {knob}, vs f64")
        mask = np.ones(po0.size, bool)
        s = sh(o_view).resolved()
        mask[(s.offset + np.arange(s.nrows)[:, None] + np.arange(s.ncols)[None, :] * s.stride).ravel()] = False
        assert np.array_equal(got[mask], po0[mask]), f"mid tile {knob} wrote outside its output view"
    finally:
        gpu.set_tuning("f32_mid", old)


def test_encased_builders_and_into_inner(gpu):
    """tensor.rs:132-173 (`build_uninit_encased`, `build_encase`) and :277-279 (`into_inner`): items of a shader struct travel in their storage layout --
    here a structured dtype with the 16-byte alignment WGSL gives a vec3 member -- and `into_inner` keeps the allocation while consuming the tensor."""
    import wgmath_amd as wg
    S = wg.BufferUsages
    item = np.dtype({"names": ["p", "id"], "formats": [(np.float32, 3), np.uint32], "offsets": [0, 12], "itemsize": 16})
    host = np.zeros(37, item)
    host["p"] = np.arange(37 * 3, dtype=np.float32).reshape(37, 3)
    host["id"] = np.arange(37, dtype=np.uint32) * 7
    t = wg.TensorBuilder.vector(37, S.STORAGE | S.COPY_SRC).build_encase(gpu.device(), host)
    assert t.bytes_len() == 37 * 16
    assert t.read_bytes(gpu.device()) == host.tobytes()
    u = wg.TensorBuilder.vector(5, S.STORAGE).build_uninit_encased(gpu.device(), item)
    assert u.bytes_len() == 80 and u.len() == 5
    with pytest.raises(AssertionError, match="Incorrect number of elements"):
        wg.TensorBuilder.vector(38, S.STORAGE).build_encase(gpu.device(), host)
    ptr = t.device_ptr()
    inner = t.into_inner()
    assert isinstance(inner, wg.GpuBuffer) and inner.size == 37 * 16 and inner.device_ptr() == ptr
    with pytest.raises(AssertionError, match="already consumed"):
        t.into_inner()
    del t  # the consumed tensor no longer owns anything: dropping it must not free the allocation
    again = wg.GpuTensor.wrap(gpu.device(), inner.device_ptr(), (37 * 4,), np.uint32, keepalive=inner)
    assert again.read_bytes(gpu.device()) == host.tobytes()
    # GpuVector::encase / uninit_encased, bytes_len_encased, copy_from_encased (tensor.rs:217-241,633-655)
    v = wg.GpuTensor.encase(gpu.device(), host, S.STORAGE | S.COPY_SRC)
    w = wg.GpuTensor.uninit_encased(gpu.device(), 37, S.STORAGE | S.COPY_SRC | S.COPY_DST, item_dtype=item)
    assert v.bytes_len_encased() == w.bytes_len_encased() == 37 * 16
    enc = gpu.device().create_command_encoder()
    w.copy_from_encased(enc, v)
    gpu.queue().submit([enc.finish()])
    assert w.read_bytes(gpu.device()) == host.tobytes()


@pytest.mark.parametrize("M,K,N", [(64, 160, 12288), (12288, 160, 64), (48, 256, 16384), (64, 132, 24576)])
@pytest.mark.parametrize("tr", [False, True])
def test_gemm_f32_few_rows_or_columns_one_tile_per_cu(gpu, oracle_c, M, K, N, tr):
    """One 64-row (or 64-column) strip of 192 .. 512 tiles of 64 x 64: the launcher's own choice there is the mid family's 64 x 64 tile UNSPLIT (gemm_f32.hip,
    few-row branch) -- against the oracle and f64, and bit for bit against the same tile forced through the knob (K % 32 != 0 in two of the cases: its remainder tile)."""
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(M + 3 * K + 5 * N + int(tr))
    a = (rng.random(M * K, dtype=np.float32) * 2 - 1).astype(np.float32)
    b = (rng.random(K * N, dtype=np.float32) * 2 - 1).astype(np.float32)
    s1 = wo.Shape(K, M, 1) if tr else wo.Shape(M, K, 1)
    s2, so = wo.Shape(K, N, 1), wo.Shape(M, N, 1)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    orc = np.zeros(M * N, np.float32)
    oracle_c.gemm(int(variant), orc, so, a, s1, b, s2)
    m1, m2 = upload(gpu, (K, M, 1) if tr else (M, K, 1), a), upload(gpu, (K, N, 1), b)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    A, B = wo.view(a, s1)[:, :, 0], wo.view(b, s2)[:, :, 0]
    truth, sabs = wo.gemm_f64(A.T if tr else A, B)
    got = {}
    for knob in (-1, 64064):
        old = gpu.set_tuning("f32_mid", knob)
        try:
            out = upload(gpu, (M, N, 1), np.full(M * N, np.nan, np.float32))
            run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, m1, m2, variant))
            got[knob] = out.read(gpu.device())
        finally:
            gpu.set_tuning("f32_mid", old)
        g = wo.view(got[knob], so)[:, :, 0]
        U.assert_close_f64(g, truth, K, sabs, f"few rows / columns, knob {knob}: gemm {M}x{K}x{N} tr={tr} vs f64")
        U.assert_close_oracle(g, wo.view(orc, so)[:, :, 0], K, sabs, f"few rows / columns, knob {knob} vs oracle")
    U.assert_bits_equal(got[-1], got[64064], "the launcher's choice is the unsplit 64 x 64 tile")


@pytest.mark.parametrize("M,K,N,mats", [(4096, 256, 4096, 1), (4104, 328, 4360, 1), (2048, 192, 2048, 4)])
@pytest.mark.parametrize("tr", [False, True])
def test_gemm_f16_short_k_on_a_large_output_is_the_256x128_tile(gpu, M, K, N, mats, tr):
    """From one round of 256 x 256 tiles on and a short K the launcher's own choice is the 256 x 128 tile with two workgroups per CU (gemm_f16.hip: the model
    of the two kernels' times per round): same bits as that tile forced, within the f16 gate of f64 -- whole, ragged (M, N not multiples of the tile, K % 64 != 0)
    and batched outputs."""
    wg, wo = _wg(), _wo()
    rng = np.random.default_rng(M * 31 + K * 17 + N + mats + int(tr))
    a = (rng.random(M * K * mats, dtype=np.float32) * 2 - 1).astype(np.float16)
    b = (rng.random(K * N * mats, dtype=np.float32) * 2 - 1).astype(np.float16)
    s1 = wo.Shape(K, M, mats) if tr else wo.Shape(M, K, mats)
    s2, so = wo.Shape(K, N, mats), wo.Shape(M, N, mats)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    m1 = upload(gpu, (K, M, mats) if tr else (M, K, mats), a, np.float16)
    m2 = upload(gpu, (K, N, mats), b, np.float16)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    got = {}
    old = gpu.get_tuning("f16_tile")
    try:
        for tile in (0, 256128, 256):
            gpu.set_tuning("f16_tile", tile)
            out = upload(gpu, (M, N, mats), np.full(M * N * mats, np.nan, np.float16), np.float16)
            run_pass(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, out, m1, m2, variant))
            got[tile] = out.read(gpu.device())
    finally:
        gpu.set_tuning("f16_tile", old)
    U.assert_bits_equal(got[0], got[256128], f"f16 {M}x{K}x{N}x{mats} tr={tr}: the launcher's choice vs the 256 x 128 tile forced")
    U.assert_bits_equal(got[0], got[256], f"f16 {M}x{K}x{N}x{mats} tr={tr}: the 256 x 128 tile vs the 256 x 256 kernel (same k order per element)")
    A, B = wo.view(a, s1), wo.view(b, s2)
    for t in range(mats):
        amk = (A[:, :, t].T if tr else A[:, :, t]).astype(np.float64)
        f16_check(wo.view(got[0], so)[:, :, t], amk, B[:, :, t].astype(np.float64), K, f"f16 gemm {M}x{K}x{N} mat {t} tr={tr}, launcher's choice")
