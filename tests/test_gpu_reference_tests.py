"""The reference's own four hot-path tests, transcribed call for call onto the mirror API, run on the HIP kernels.

  gpu_gemm       crates/wgebra/src/linalg/gemm.rs:141-202
  gpu_gemv       crates/wgebra/src/linalg/gemv.rs:152-197
  gpu_reduce     crates/wgebra/src/linalg/reduce.rs:136-179
  gpu_op_assign  crates/wgebra/src/linalg/op_assign.rs:108-157

Same shapes, same input distributions (U[0,1) where the reference uses `new_random`, seeded here), same CPU check
(what nalgebra computes), same literal tolerance.  Each test then tightens the bar: vs f64 with the K-scaled gate, and
vs the oracle (bit-exact where the order is specified).
"""
import numpy as np
import pytest

import _util as U

pytestmark = pytest.mark.gpu


def test_gpu_gemm(gpu, oracle_c):
    import wgmath_amd as wg
    from oracle import wgsl_oracle as wo

    gemm = wg.Gemm.from_device(gpu.device())
    shapes = wg.ViewShapeBuffers.new()
    NROWS = NCOLS = 256
    rng = np.random.default_rng(20251205)
    m1_cpu = rng.random((NROWS, NCOLS), dtype=np.float32)  # DMatrix::<f32>::new_random
    m2_cpu = rng.random((NCOLS, NROWS), dtype=np.float32)
    lhs_cpu = np.zeros((NROWS, NROWS), np.float32)

    S = wg.BufferUsages
    m1 = wg.TensorBuilder.matrix(NROWS, NCOLS, S.STORAGE).build_init(gpu.device(), m1_cpu)
    m2 = wg.TensorBuilder.matrix(NCOLS, NROWS, S.STORAGE).build_init(gpu.device(), m2_cpu)
    result = wg.TensorBuilder.matrix(NROWS, NROWS, S.STORAGE | S.COPY_SRC).build_init(gpu.device(), lhs_cpu)
    staging = wg.TensorBuilder.matrix(NROWS, NROWS, S.MAP_READ | S.COPY_DST).build(gpu.device())

    V = wg.GemmVariant
    for variant in [V.Gemm, V.GemmTr, V.GemmFast, V.GemmTrFast]:
        encoder = gpu.device().create_command_encoder()
        pass_ = encoder.compute_pass("test", None)
        gemm.dispatch_generic(gpu.device(), shapes, pass_, result.as_embedded_view(), m1.as_embedded_view(),
                              m2.as_embedded_view(), variant)
        pass_.end()
        staging.copy_from(encoder, result)
        gpu.queue().submit([encoder.finish()])
        gpu_result = staging.read(gpu.device()).reshape(NROWS, NROWS, order="F")

        tr = variant in (V.GemmTr, V.GemmTrFast)
        a = m1_cpu.T if tr else m1_cpu
        cpu_result = a @ m2_cpu  # &m1_cpu * &m2_cpu / tr_mul
        assert U.relative_eq(gpu_result, cpu_result, epsilon=U.REF_ABS_EPS), variant  # the reference's bar

        truth, sabs = wo.gemm_f64(a, m2_cpu)
        U.assert_close_f64(gpu_result, truth, NCOLS, sabs, f"gemm {variant!r} vs f64")
        orc = np.zeros(NROWS * NROWS, np.float32)
        flat1, flat2 = m1_cpu.reshape(-1, order="F"), m2_cpu.reshape(-1, order="F")
        oracle_c.gemm(int(variant), orc, wo.Shape(NROWS, NROWS), flat1, wo.Shape(NROWS, NCOLS), flat2, wo.Shape(NCOLS, NROWS))
        U.assert_close_oracle(gpu_result.reshape(-1, order="F"), orc, NCOLS, sabs.reshape(-1, order="F"), f"gemm {variant!r} vs oracle")


def test_gpu_gemv(gpu, oracle_c):
    import wgmath_amd as wg
    from oracle import wgsl_oracle as wo

    gemv = wg.Gemv.from_device(gpu.device())
    shapes = wg.ViewShapeBuffers.new()
    NROWS = NCOLS = 1024
    rng = np.random.default_rng(20251206)
    m_cpu = rng.random((NROWS, NCOLS), dtype=np.float32)
    v_cpu = rng.random(NCOLS, dtype=np.float32)
    lhs_cpu = rng.random(NROWS, dtype=np.float32)  # random pre-fill: the kernel must overwrite all of it

    S = wg.BufferUsages
    m = wg.TensorBuilder.matrix(NROWS, NCOLS, S.STORAGE).build_init(gpu.device(), m_cpu)
    v = wg.TensorBuilder.vector(NCOLS, S.STORAGE).build_init(gpu.device(), v_cpu)
    result = wg.TensorBuilder.vector(NROWS, S.STORAGE | S.COPY_SRC).build_init(gpu.device(), lhs_cpu)
    staging = wg.TensorBuilder.vector(NROWS, S.MAP_READ | S.COPY_DST).build(gpu.device())

    V = wg.GemvVariant
    for variant in [V.Gemv, V.GemvTr, V.GemvFast, V.GemvTrFast]:
        encoder = gpu.device().create_command_encoder()
        pass_ = encoder.compute_pass("test", None)
        gemv.dispatch_generic(gpu.device(), shapes, pass_, result, m, v, variant)
        pass_.end()
        staging.copy_from(encoder, result)
        gpu.queue().submit([encoder.finish()])
        gpu_result = staging.read(gpu.device())

        tr = variant in (V.GemvTr, V.GemvTrFast)
        a = m_cpu.T if tr else m_cpu
        cpu_result = a @ v_cpu
        assert U.relative_eq(gpu_result, cpu_result, epsilon=U.REF_ABS_EPS), variant

        truth, sabs = wo.gemm_f64(a, v_cpu[:, None])
        U.assert_close_f64(gpu_result, truth, NCOLS, sabs, f"gemv {variant!r} vs f64")
        orc = lhs_cpu.copy()
        oracle_c.gemv(int(variant), orc, wo.Shape(NROWS), m_cpu.reshape(-1, order="F"), wo.Shape(NROWS, NCOLS), v_cpu, wo.Shape(NCOLS))
        U.assert_close_oracle(gpu_result, orc, NCOLS, sabs, f"gemv {variant!r} vs oracle")


def test_gpu_reduce(gpu, oracle_c):
    import wgmath_amd as wg
    from oracle import wgsl_oracle as wo

    shapes = wg.ViewShapeBuffers.new()
    R = wg.ReduceOp
    rng = np.random.default_rng(20251207)
    S = wg.BufferUsages
    for op in [R.Min, R.Max, R.Sum, R.SqNorm, R.Prod]:
        reduce = wg.Reduce.new(gpu.device(), op)
        encoder = gpu.device().create_command_encoder()
        LEN = 345
        numbers = rng.random(LEN, dtype=np.float32)
        vector = wg.TensorBuilder.vector(LEN, S.STORAGE).build_init(gpu.device(), numbers)
        result = wg.TensorBuilder.scalar(S.STORAGE | S.COPY_SRC).build(gpu.device())
        staging = wg.TensorBuilder.scalar(S.MAP_READ | S.COPY_DST).build(gpu.device())
        pass_ = encoder.compute_pass("test", None)
        reduce.dispatch(gpu.device(), shapes, pass_, vector, result)
        pass_.end()
        staging.copy_from(encoder, result)
        gpu.queue().submit([encoder.finish()])
        got = staging.read(gpu.device())[0]
        assert U.relative_eq(got, reduce.eval_cpu(numbers), epsilon=U.REF_ABS_EPS), op  # the reference's bar
        expected = oracle_c.reduce(int(op), numbers, wo.Shape(LEN))
        U.assert_bits_equal(np.float32(got), np.float32(expected), f"reduce {op!r} vs oracle (bit-exact order)")


def test_gpu_op_assign(gpu, oracle_c):
    import wgmath_amd as wg
    from oracle import wgsl_oracle as wo

    O = wg.OpAssignVariant
    shapes = wg.ViewShapeBuffers.new()
    S = wg.BufferUsages
    ref = U.golden("op_assign_ref_1757")
    for op in [O.Add, O.Sub, O.Mul, O.Div, O.Copy]:  # the reference leaves Copy untested; covered here
        op_assign = wg.OpAssign.new(gpu.device(), op)
        encoder = gpu.device().create_command_encoder()
        LEN = 1757
        v0 = (np.arange(LEN, dtype=np.float32) + np.float32(0.1)).astype(np.float32)
        v1 = (np.arange(LEN, dtype=np.float32) * np.float32(10.0) + np.float32(0.1)).astype(np.float32)
        gpu_v0 = wg.TensorBuilder.vector(LEN, S.STORAGE | S.COPY_SRC).build_init(gpu.device(), v0)
        gpu_v1 = wg.TensorBuilder.vector(LEN, S.STORAGE).build_init(gpu.device(), v1)
        staging = wg.TensorBuilder.vector(LEN, S.MAP_READ | S.COPY_DST).build(gpu.device())
        pass_ = encoder.compute_pass("test", None)
        op_assign.dispatch(gpu.device(), shapes, pass_, gpu_v0, gpu_v1)
        pass_.end()
        staging.copy_from(encoder, gpu_v0)
        gpu.queue().submit([encoder.finish()])
        got = staging.read(gpu.device())
        with np.errstate(all="ignore"):
            cpu_result = {O.Add: v0 + v1, O.Sub: v0 - v1, O.Mul: v0 * v1, O.Div: v0 / v1, O.Copy: v1.copy()}[op]
        assert U.relative_eq(got, cpu_result, epsilon=1.0e-7), op  # the reference's bar
        U.assert_bits_equal(got, ref[f"expected_{int(op)}"], f"op_assign {op!r} vs the committed known-answer vector")
        orc = v0.copy()
        oracle_c.op_assign(int(op), orc, wo.Shape(LEN), v1, wo.Shape(LEN))
        U.assert_bits_equal(got, orc, f"op_assign {op!r} vs oracle")
