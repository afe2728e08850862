"""BASELINE-size runs (configs 2-4): sizes the oracle cannot finish in seconds, checked through sampled f64 truth and
size-independent properties (row-sum identity through an independent kernel, linearity, bit-exact samples)."""
import numpy as np
import pytest

import _util as U

pytestmark = pytest.mark.gpu


def _wg():
    import wgmath_amd as wg
    return wg


S_ALL = 128 | 4 | 8


def up(gpu, shape, flat, dtype=np.float32):
    wg = _wg()
    return wg.TensorBuilder.tensor(shape, S_ALL).build_init(gpu.device(), np.asarray(flat, dtype), dtype)


def run(gpu, fn):
    enc = gpu.device().create_command_encoder()
    with enc.compute_pass("full", None) as p:
        fn(p)
    gpu.queue().submit([enc.finish()])


def rnd(seed, n, dtype=np.float32):
    rng = np.random.default_rng(seed)
    return (rng.random(n, dtype=np.float32) * 2 - 1).astype(dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,M,N,K,tr", [(np.float32, 32000, 16, 4096, False), (np.float32, 32000, 16, 4096, True), (np.float32, 65536, 8, 4096, False),
                                            (np.float16, 65536, 8, 4096, True)])
def test_few_columns_on_a_matrix_past_the_infinity_cache(gpu, dtype, M, N, K, tr):
    """The few-column kernel (gemm_f32_skinny.hip) reads a streamed operand of 384 MiB and more with the non-temporal hint on its LDS-DMA pieces (SkinnyArgs::a_nt:
    bench workload gemm_f32_fewcols_32000x16x4096 0.73 -> 0.80 of HBM peak): those launches at full size, every element written, sampled rows against f64 with
    the kernel's usual bound (the smaller parity shapes of test_gpu_parity.py all run without the hint)."""
    wg = _wg()
    rng = np.random.default_rng(M + N + K + int(tr))
    a = (rng.random(M * K, dtype=np.float32) * 2 - 1).astype(dtype)
    b = (rng.random(K * N, dtype=np.float32) * 2 - 1).astype(dtype)
    assert a.nbytes >= 384 << 20
    ta = up(gpu, (K, M) if tr else (M, K), a, dtype)
    tb = up(gpu, (K, N), b, dtype)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    whole = up(gpu, (M, N), np.full(M * N, np.nan, dtype), dtype)
    run(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, whole, ta, tb, variant))
    got = whole.read(gpu.device()).reshape(M, N, order="F")
    assert not np.isnan(got.astype(np.float32)).any()
    A = (a.reshape(K, M, order="F").T if tr else a.reshape(M, K, order="F"))
    B = b.reshape(K, N, order="F").astype(np.float64)
    idx = np.unique(rng.integers(0, M, 96))
    exact = A[idx].astype(np.float64) @ B
    sabs = np.abs(A[idx].astype(np.float64)) @ np.abs(B)
    if dtype == np.float32:
        U.assert_close_f64(got[idx], exact, K, sabs, f"few columns {M}x{N}x{K} tr={tr} sampled vs f64")
    else:
        assert (np.abs(got[idx].astype(np.float64) - exact) <= np.abs(exact) * 2.0 ** -11 + np.sqrt(K) * 2.0 ** -23 * sabs + 2.0 ** -24).all()


@pytest.mark.parametrize("tr", [False, True])
def test_config2_gemm_f32_4096(gpu, tr):
    wg = _wg()
    n = 4096
    a, b = rnd(1, n * n), rnd(2, n * n)
    ta, tb = up(gpu, (n, n), a), up(gpu, (n, n), b)
    tc = up(gpu, (n, n), np.zeros(n * n, np.float32))
    gemm, gemv, shapes = wg.Gemm.from_device(gpu.device()), wg.Gemv.from_device(gpu.device()), wg.ViewShapeBuffers()
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    run(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, tc, ta, tb, variant))
    C = tc.read(gpu.device()).reshape(n, n, order="F")
    A = a.reshape(n, n, order="F")
    A = A.T if tr else A
    B = b.reshape(n, n, order="F")
    rows = np.unique(np.random.default_rng(3).integers(0, n, 48))
    a64, b64 = A[rows].astype(np.float64), B.astype(np.float64)
    U.assert_close_f64(C[rows], a64 @ b64, n, np.abs(a64) @ np.abs(b64), f"gemm f32 4096^3 tr={tr}: sampled rows vs f64")
    # checksum of checksums through independent kernels: C.1 == op(A).(B.1)   (Gemv N twice, or Gemv T for op(A) = A^T)
    ones = up(gpu, (n,), np.ones(n, np.float32))
    b1, ab1, c1 = up(gpu, (n,), np.zeros(n, np.float32)), up(gpu, (n,), np.zeros(n, np.float32)), up(gpu, (n,), np.zeros(n, np.float32))
    run(gpu, lambda p: (gemv.dispatch(gpu.device(), shapes, p, b1, tb, ones),
                        gemv.dispatch_generic(gpu.device(), shapes, p, ab1, ta, b1, wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv),
                        gemv.dispatch(gpu.device(), shapes, p, c1, tc, ones)))
    lhs, rhs = c1.read(gpu.device()).astype(np.float64), ab1.read(gpu.device()).astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ (np.abs(B).astype(np.float64) @ np.ones(n))
    assert (np.abs(lhs - rhs) <= 8 * np.sqrt(n) * 2.0 ** -24 * scale).all(), "row-sum identity C.1 == A.(B.1) violated"


def test_config3_gemm_f16_8192(gpu):
    wg = _wg()
    n = 8192
    a, b = rnd(4, n * n, np.float16), rnd(5, n * n, np.float16)
    ta, tb = up(gpu, (n, n), a, np.float16), up(gpu, (n, n), b, np.float16)
    tc = up(gpu, (n, n), np.zeros(n * n, np.float16), np.float16)
    gemm, shapes = wg.Gemm.from_device(gpu.device()), wg.ViewShapeBuffers()
    for variant in (wg.GemmVariant.Gemm, wg.GemmVariant.GemmTr):
        run(gpu, lambda p: gemm.dispatch_generic(gpu.device(), shapes, p, tc, ta, tb, variant))
        C = tc.read(gpu.device()).reshape(n, n, order="F")
        A = a.reshape(n, n, order="F")
        A = A.T if variant == wg.GemmVariant.GemmTr else A
        B = b.reshape(n, n, order="F")
        rng = np.random.default_rng(6)
        rows, cols = np.unique(rng.integers(0, n, 40)), np.unique(rng.integers(0, n, 512))
        a64, b64 = A[rows].astype(np.float64), B[:, cols].astype(np.float64)
        truth, sabs = a64 @ b64, np.abs(a64) @ np.abs(b64)
        tol = U.f32_gate(n, sabs) + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25
        err = np.abs(C[np.ix_(rows, cols)].astype(np.float64) - truth)
        assert (err <= tol).all(), f"f16 gemm 8192^3 {variant!r}: worst err/tol {(err / tol).max():.3g}"
        assert np.isfinite(C).all()


def test_config4_gemv_4096x65536_and_transpose(gpu):
    wg = _wg()
    R, Cn = 4096, 65536
    m = rnd(7, R * Cn)
    tm = up(gpu, (R, Cn), m)
    M = m.reshape(R, Cn, order="F")
    gemv, shapes = wg.Gemv.from_device(gpu.device()), wg.ViewShapeBuffers()
    for tr in (False, True):
        vlen, olen = (R, Cn) if tr else (Cn, R)
        v1, v2 = rnd(8 + tr, vlen), rnd(10 + tr, vlen)
        tv1, tv2, tv12 = up(gpu, (vlen,), v1), up(gpu, (vlen,), v2), up(gpu, (vlen,), v1 + v2)
        o1, o2, o12 = (up(gpu, (olen,), np.full(olen, np.nan, np.float32)) for _ in range(3))
        variant = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv
        run(gpu, lambda p: [gemv.dispatch_generic(gpu.device(), shapes, p, o, tm, v, variant) for o, v in ((o1, tv1), (o2, tv2), (o12, tv12))])
        g1, g2, g12 = (o.read(gpu.device()).astype(np.float64) for o in (o1, o2, o12))
        idx = np.unique(np.random.default_rng(12).integers(0, olen, 96))
        a64 = (M[:, idx].T if tr else M[idx, :]).astype(np.float64)
        U.assert_close_f64(g1[idx], a64 @ v1.astype(np.float64), vlen, np.abs(a64) @ np.abs(v1).astype(np.float64), f"gemv tr={tr} sampled vs f64")
        # linearity on every output: m(v1+v2) == m v1 + m v2 up to rounding (v1+v2 is rounded once on the host)
        sabs_bound = np.sqrt(vlen) * 8 * 2.0 ** -24 * (np.abs(g1) + np.abs(g2) + 1.0) * np.sqrt(vlen)
        assert (np.abs(g12 - (g1 + g2)) <= sabs_bound).all(), f"gemv tr={tr}: linearity violated"
        # multi-RHS form (out_ncols = 8) reproduces the single-RHS result bit for bit (same kernel order per column? no: same bits required only per launch config)
    # 8 RHS columns in one dispatch: each column within tolerance of f64 on sampled rows
    V = rnd(20, Cn * 8)
    tV, tO = up(gpu, (Cn, 8), V), up(gpu, (R, 8), np.zeros(R * 8, np.float32))
    run(gpu, lambda p: gemv.dispatch(gpu.device(), shapes, p, tO, tm, tV))
    O = tO.read(gpu.device()).reshape(R, 8, order="F")
    idx = np.unique(np.random.default_rng(13).integers(0, R, 64))
    a64, v64 = M[idx].astype(np.float64), V.reshape(Cn, 8, order="F").astype(np.float64)
    U.assert_close_f64(O[idx], a64 @ v64, Cn, np.abs(a64) @ np.abs(v64), "gemv 8 RHS sampled vs f64")


def test_config4_batched_reduce_4096x65536(gpu, oracle_c):
    wg = _wg()
    from oracle import wgsl_oracle as wo
    n, nvec = 65536, 4096
    x = rnd(30, n * nvec)
    tx = up(gpu, (n, nvec), x)
    X = x.reshape(n, nvec, order="F")
    shapes = wg.ViewShapeBuffers()
    for op in wg.ReduceOp:
        red = wg.Reduce.new(gpu.device(), op)
        res = up(gpu, (nvec,), np.full(nvec, np.nan, np.float32))
        run(gpu, lambda p: red.dispatch_batched(gpu.device(), shapes, p, tx, res))
        got = res.read(gpu.device())
        # a sample of vectors is bit-identical to the reference order ...
        for c in (0, 1, 777, 2048, 4095):
            exp = oracle_c.reduce(int(op), x, wo.Shape(n, 1, 1, 1, 1, c * n))
            U.assert_bits_equal(got[c:c + 1], np.array([exp], np.float32), f"batched reduce {op!r} vector {c}")
        # ... and every vector is within the rounding bound of the f64 result
        x64 = X.astype(np.float64)
        if op == wg.ReduceOp.Sum:
            assert (np.abs(got - x64.sum(0)) <= n * 2.0 ** -24 * np.abs(x64).sum(0)).all()
        elif op == wg.ReduceOp.SqNorm:
            assert (np.abs(got - (x64 * x64).sum(0)) <= n * 2.0 ** -23 * (x64 * x64).sum(0)).all()
        elif op == wg.ReduceOp.Min:
            assert np.array_equal(got, X.min(0))
        elif op == wg.ReduceOp.Max:
            assert np.array_equal(got, X.max(0))


def test_op_assign_256M(gpu):
    wg = _wg()
    n = 1 << 28
    a, b = rnd(40, n), rnd(41, n)
    ta, tb = up(gpu, (n,), a), up(gpu, (n,), b)
    shapes = wg.ViewShapeBuffers()
    run(gpu, lambda p: wg.OpAssign.new(gpu.device(), wg.OpAssignVariant.Mul).dispatch(gpu.device(), shapes, p, ta, tb))
    U.assert_bits_equal(ta.read(gpu.device()), a * b, "op_assign Mul 2^28")
    run(gpu, lambda p: wg.OpAssign.new(gpu.device(), wg.OpAssignVariant.Copy).dispatch(gpu.device(), shapes, p, ta, tb))
    U.assert_bits_equal(ta.read(gpu.device()), b, "op_assign Copy 2^28")


def test_addressing_beyond_4GiB(gpu):
    """An 8 GiB f32 matrix (32768 x 65536: 2^31 elements, > the reference's 600 MB buffer cap and > 2^32 bytes): GEMV N/T and a
    batched Reduce must address it with 64-bit arithmetic. The matrix is a 2^24-element random block tiled on the device, so the
    expectation is computable on the host: column c equals block column c % 512."""
    wg = _wg()
    from wgmath_amd._lib import check, lib
    R, C, BLK = 32768, 65536, 1 << 24
    blk = rnd(77, BLK)
    S = wg.BufferUsages
    tm = wg.TensorBuilder.matrix(R, C, S.STORAGE | S.COPY_DST).build(gpu.device())
    tb = up(gpu, (BLK,), blk)
    for off in range(0, R * C, BLK):
        check(lib.wg_buf_copy(gpu._ctx.handle, tb._h, 0, tm._h, off * 4, BLK * 4))
    Mb = blk.reshape(R, BLK // R, order="F").astype(np.float64)  # R x 512: the distinct columns
    reps = C // (BLK // R)
    gemv, shapes = wg.Gemv.from_device(gpu.device()), wg.ViewShapeBuffers()
    v = rnd(78, C)
    tv, to = up(gpu, (C,), v), up(gpu, (R,), np.zeros(R, np.float32))
    run(gpu, lambda p: gemv.dispatch(gpu.device(), shapes, p, to, tm, tv))
    got = to.read(gpu.device()).astype(np.float64)
    v64 = v.astype(np.float64).reshape(reps, BLK // R)
    truth = Mb @ v64.sum(0)
    sabs = np.abs(Mb) @ np.abs(v64).sum(0)
    U.assert_close_f64(got, truth, C, sabs, "8 GiB gemv N")
    w = rnd(79, R)
    tw, to2 = up(gpu, (R,), w), up(gpu, (C,), np.zeros(C, np.float32))
    run(gpu, lambda p: gemv.dispatch_tr(gpu.device(), shapes, p, to2, tm, tw))
    got2 = to2.read(gpu.device()).astype(np.float64)
    col = Mb.T @ w.astype(np.float64)  # 512 distinct dot products
    U.assert_close_f64(got2, np.tile(col, reps), R, np.tile(np.abs(Mb).T @ np.abs(w).astype(np.float64), reps), "8 GiB gemv T")
    red = wg.Reduce.new(gpu.device(), wg.ReduceOp.Sum)
    res = up(gpu, (C,), np.zeros(C, np.float32))
    run(gpu, lambda p: red.dispatch_batched(gpu.device(), shapes, p, tm, res))
    sums = res.read(gpu.device())
    assert np.array_equal(sums, np.tile(sums[:BLK // R], reps)), "columns past 4 GiB do not repeat the block's sums bit for bit"
    assert np.allclose(sums[:BLK // R], Mb.sum(0), rtol=0, atol=R * 2.0 ** -24 * np.abs(Mb).sum(0).max())


# --------------------------------------------------------------------------------------------------------
# config 5: f16 GEMM 32768^3 through the M-sharded entry point (wg_gemm_sharded). One GPU here, so one rank -- but the whole
# path: a real RCCL communicator created through the C ABI, staging cube + ncclAllGather + cube_to_matrix relayout per N-panel
# (WG_GATHER_RCCL), on the unmasked 256-CU stream and on the 224-CU masked stream the multi-rank bench gives that engine, and the
# panel-wise strided-output path straight into C (WG_GATHER_NONE). >= 64 sampled rows x 512 columns against f64.
# --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("engine", ["rccl", "staged"])
@pytest.mark.parametrize("cus", [None, 248, -248])  # -248: the 8 missing CUs all from one XCD (wg_ctx_create_with_cu_count_one_xcd: what bench.py gives the RCCL engine)
@pytest.mark.parametrize("M,K,N,panel,bitwise", [(4096, 1024, 4096, 1024, True), (4096, 512 + 32, 8192, 4096, True), (8192, 512, 8192, 2048, True),
                                                 (4096, 512 + 32, 4352, 1024, False), (2048, 768, 8448, 512, False)])
def test_sharded_gemm_one_launch_is_bit_identical_to_panel_launches(engine, cus, M, K, N, panel, bitwise):
    """wg_gemm_sharded's one-launch-per-step form (the rank's whole f16 product as ONE kernel that walks the N-panels, writes them through
    to memory and raises a flag per panel; the panel's exchange waits on the flag with hipStreamWaitValue32, the relayouts follow) against
    the panel-by-panel launches: the same tiles, the same accumulation chains -- bit for bit, ragged last panel and K remainder included,
    on the full chip and on a 248-CU masked stream (31 CUs per XCD: viable now that nothing assumes 32), RCCL (1 rank) and staged engine.
    (`bitwise` False: the panel launches of that shape are split along K -- few tiles per panel with a K remainder, or a launcher
    estimate -- i.e. another summation order: both forms are then held to the f64 bound, and the one-launch form to itself.)"""
    import os
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    wg = _wg()
    from wgmath_amd.sharded import Comm, GatherMode, new_unique_id
    inst = wg.GpuInstance.new(0, cu_count=abs(cus), one_xcd=cus < 0) if cus else wg.GpuInstance.new(0)
    dev = inst.device()
    comm = Comm(inst, 1, 0, new_unique_id() if engine == "rccl" else None)
    rng = np.random.default_rng(M + K + N + panel)
    a = (rng.random(M * K, dtype=np.float32) * 2 - 1).astype(np.float16)
    b = (rng.random(K * N, dtype=np.float32) * 2 - 1).astype(np.float16)
    A = wg.TensorBuilder.matrix(M, K, S_ALL).build_init(dev, a)
    B = wg.TensorBuilder.matrix(K, N, S_ALL).build_init(dev, b)
    mode = GatherMode.RCCL if engine == "rccl" else GatherMode.PEER_STAGED
    if engine == "staged":
        comm.stage_reserve(2 * M * N * 2)
    res = []
    for one in (False, True, True):
        comm.set_one_launch(one)
        C = wg.TensorBuilder.matrix(M, N, S_ALL).build_init(dev, np.full(M * N, np.nan, np.float16))
        comm.sharded_gemm(C, A, B, 0, mode, panel)
        comm.join()
        inst.sync()
        res.append(C.read(dev).view(np.uint16).copy())
    assert not np.isnan(res[0].view(np.float16)).any()
    assert np.array_equal(res[2], res[1])
    if bitwise and cus is None:  # (on 248 CUs the panel launches of these shapes cut their last 8 tiles along K: another order)
        assert np.array_equal(res[1], res[0])
    A64, B64 = a.reshape(K, M).T.astype(np.float64), b.reshape(N, K).T.astype(np.float64)
    rows = np.unique(rng.integers(0, M, 96))
    truth, sabs = A64[rows] @ B64, np.abs(A64[rows]) @ np.abs(B64)
    tol = U.f32_gate(K, sabs) + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25
    for r in res[:2]:
        got = r.view(np.float16).reshape(N, M).T.astype(np.float64)
        assert (np.abs(got[rows] - truth) <= tol).all()
    comm.close()
    inst.close()


@pytest.mark.parametrize("engine,cus", [("rccl", None), ("staged", None), ("rccl", -248)])  # -248: the RCCL engine's own stream in bench.py (8 CUs of one XCD left out)
@pytest.mark.parametrize("M,K,N,widths,one_launch_shape", [
    (8192, 512, 8192, (2048, 2048, 1536, 1024, 768, 512, 256), True),            # two equal panels, then 6, 4, 3, 2, 1 tile columns
    (4096, 512 + 32, 8192 + 128, (2048, 2048, 2048, 1024, 512, 256, 384), True),  # K remainder; N not a multiple of the tile: the ragged rest in the last panel
    (8192, 512, 8192, (4096, 2048, 1024, 512, 256, 256), True),                   # ONE main panel and a tail of five
    (4096, 512, 8192, (1024, 2048, 2048, 3072), False),                           # not "equal panels, then a tail": runs panel by panel
])
def test_sharded_gemm_tapered_tail_is_bit_identical_to_uniform_panels(engine, cus, M, K, N, widths, one_launch_shape):
    """wg_gemm_sharded_panels: the N-panels' widths given one by one. A tapered tail changes WHEN a tile's columns are exchanged, never how a
    tile is computed: in the one-launch form the result must equal, bit for bit, the one-launch result of a uniform split (same tiles, same
    accumulation chains) -- slot layout of the staging cube, per-panel counters and relayouts all follow the ragged plan. A list that is not of
    the one-launch shape runs panel by panel and is held to the f64 bound."""
    import os
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    wg = _wg()
    from wgmath_amd.sharded import Comm, GatherMode, new_unique_id
    inst = wg.GpuInstance.new(0, cu_count=abs(cus), one_xcd=cus < 0) if cus else wg.GpuInstance.new(0)
    dev = inst.device()
    comm = Comm(inst, 1, 0, new_unique_id() if engine == "rccl" else None)
    rng = np.random.default_rng(M + K + N + len(widths))
    a = (rng.random(M * K, dtype=np.float32) * 2 - 1).astype(np.float16)
    b = (rng.random(K * N, dtype=np.float32) * 2 - 1).astype(np.float16)
    A = wg.TensorBuilder.matrix(M, K, S_ALL).build_init(dev, a)
    B = wg.TensorBuilder.matrix(K, N, S_ALL).build_init(dev, b)
    mode = GatherMode.RCCL if engine == "rccl" else GatherMode.PEER_STAGED
    if engine == "staged":
        comm.stage_reserve(2 * M * N * 2)
    comm.set_one_launch(True)
    res = []
    for pc in (2048, list(widths), list(widths)):
        C = wg.TensorBuilder.matrix(M, N, S_ALL).build_init(dev, np.full(M * N, np.nan, np.float16))
        comm.sharded_gemm(C, A, B, 0, mode, pc)
        comm.join()
        inst.sync()
        res.append(C.read(dev).view(np.uint16).copy())
    assert not np.isnan(res[1].view(np.float16)).any(), "unwritten elements"
    assert np.array_equal(res[1], res[2])
    if one_launch_shape:
        assert np.array_equal(res[0], res[1]), "the tapered one-launch result differs from the uniform one-launch result"
    A64, B64 = a.reshape(K, M).T.astype(np.float64), b.reshape(N, K).T.astype(np.float64)
    rows = np.unique(rng.integers(0, M, 96))
    truth, sabs = A64[rows] @ B64, np.abs(A64[rows]) @ np.abs(B64)
    tol = U.f32_gate(K, sabs) + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25
    got = res[1].view(np.float16).reshape(N, M).T.astype(np.float64)
    assert (np.abs(got[rows] - truth) <= tol).all()
    # bad lists are refused with the reference-style precondition error, not run
    for bad in ([2048, 2048], [N - 2, 2], [0, N]):
        with pytest.raises(wg.WgError):
            comm.sharded_gemm(C, A, B, 0, mode, bad)
    comm.close()
    inst.close()


@pytest.mark.parametrize("engine", ["rccl", "staged"])
def test_sharded_gemm_pipelined_steps_one_launch(engine):
    """Pipelined steps of the one-launch form (each call's last panel completes behind the NEXT call's kernel; two staging cubes by step
    parity), RCCL engine (1 rank) and staged engine: five back-to-back steps on different B and different outputs with no synchronisation
    in between must each equal the unpipelined panel-by-panel result of the same operands."""
    import os
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    wg = _wg()
    from wgmath_amd.sharded import Comm, GatherMode, new_unique_id
    inst = wg.GpuInstance.new(0)
    dev = inst.device()
    comm = Comm(inst, 1, 0, new_unique_id() if engine == "rccl" else None)
    M, K, N, panel = 8192, 512, 8192, 2048  # 256 tiles per panel: the panel launches are unsplit too (same accumulation chains)
    rng = np.random.default_rng(77)
    A = wg.TensorBuilder.matrix(M, K, S_ALL).build_init(dev, (rng.random(M * K, dtype=np.float32) * 2 - 1).astype(np.float16))
    Bs = [wg.TensorBuilder.matrix(K, N, S_ALL).build_init(dev, (rng.random(K * N, dtype=np.float32) * 2 - 1).astype(np.float16)) for _ in range(5)]
    mode = GatherMode.RCCL if engine == "rccl" else GatherMode.PEER_STAGED
    if engine == "staged":
        comm.stage_reserve(2 * M * N * 2)
    res = {}
    for one, pipe in ((False, False), (True, True)):
        comm.set_one_launch(one)
        comm.set_pipelined(pipe)
        Cs = [wg.TensorBuilder.matrix(M, N, S_ALL).build_init(dev, np.full(M * N, np.nan, np.float16)) for _ in Bs]
        for B, C in zip(Bs, Cs):
            comm.sharded_gemm(C, A, B, 0, mode, panel)
        comm.join()
        inst.sync()
        res[one] = [C.read(dev).view(np.uint16).copy() for C in Cs]
    for i, (x, y) in enumerate(zip(res[False], res[True])):
        assert not np.isnan(y.view(np.float16)).any(), f"step {i}: unwritten elements"
        assert np.array_equal(x, y), f"step {i}: pipelined one-launch result differs from the panel launches"
    comm.close()
    inst.close()


@pytest.mark.parametrize("engine,cus", [("rccl", None), ("rccl", 224), ("rccl", 248), ("none", None)])
def test_config5_gemm_f16_32768_sharded_entry_point(engine, cus):
    import os
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import bench
    wg = _wg()
    from wgmath_amd.sharded import Comm, GatherMode, new_unique_id
    from wgmath_amd._lib import check, lib
    n = 32768
    inst = wg.GpuInstance.new(0, cu_count=cus) if cus else wg.GpuInstance.new(0)
    dev = inst.device()
    comm = Comm(inst, 1, 0, new_unique_id())
    assert comm.has_collectives
    A = bench.device_random(wg, inst, (n, n), np.float16, 0xA000)
    B = bench.device_random(wg, inst, (n, n), np.float16, 0xB000)
    C = wg.TensorBuilder.matrix(n, n, S_ALL).build(dev, np.float16)
    panel = bench.plan_panel_cols(n, n, cus or 256)
    mode = GatherMode.RCCL if engine == "rccl" else GatherMode.NONE  # NONE: the panel-wise strided-output path (one rank: all of C)
    comm.sharded_gemm(C, A, B, 0, mode, panel)
    comm.barrier()  # flush + 1-element all-reduce joined into the stream (what separates two steps of the bench)
    inst.sync()
    rng = np.random.default_rng(5)
    rows = np.unique(rng.integers(0, n, 80))[:72]
    cols = np.unique(rng.integers(0, n, 600))[:512]
    assert rows.size >= 64 and cols.size == 512

    def read_range(t, start, count):
        out = np.empty(count, np.float16)
        check(lib.wg_buf_read(inst._ctx.handle, t._h, start * 2, out.ctypes.data, count * 2))
        return out

    # rows of A: the operand is one 16 Mi-element block tiled (bench.device_random), column k starts at k*n
    blk = bench.rand_block(0xA000, 1 << 24, np.float16)
    a_rows = np.empty((rows.size, n), np.float64)
    for k in range(n):
        a_rows[:, k] = blk[(k * n + rows) % (1 << 24)]
    bc = np.stack([read_range(B, int(c) * n, n) for c in cols], axis=1).astype(np.float64)
    got = np.stack([read_range(C, int(c) * n, n)[rows] for c in cols], axis=1).astype(np.float64)
    truth, sabs = a_rows @ bc, np.abs(a_rows) @ np.abs(bc)
    tol = U.f32_gate(n, sabs) + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25
    err = np.abs(got - truth)
    assert (err <= tol).all(), f"config 5 via {engine} (CUs {cus}): worst err/tol {(err / tol).max():.3g}"
    # spot-check that device_random really is the tiled block the host side assumed
    assert np.array_equal(read_range(A, 5 * n + 100, 64), blk[(5 * n + 100) % (1 << 24):][:64])
    comm.close()
    del A, B, C
    inst.close()
