"""Host-side mirror of wgebra's dense operator surface over the C ABI (include/wgebra_hip.h).

  Gemm / GemmVariant         crates/wgebra/src/linalg/gemm.rs:9-127
  Gemv / GemvVariant         crates/wgebra/src/linalg/gemv.rs:9-137
  Reduce / ReduceOp          crates/wgebra/src/linalg/reduce.rs:13-113
  OpAssign / OpAssignVariant crates/wgebra/src/linalg/op_assign.rs:12-95

Signatures keep the reference's parameter order -- dispatch(device, shapes, pass, out, a, b[, variant]) -- so that a test
written against the reference reads the same here.  `device` and `shapes` are accepted for signature compatibility; the
work is enqueued on the stream behind `pass`.  Where the reference panics (assert_eq!), these raise an exception that
is also an AssertionError, carrying the reference's message.
"""
from __future__ import annotations

import enum

import numpy as np

from ._lib import check, lib
from .wgcore import ComputePass, GpuTensor, GpuTensorView, ViewShapeBuffers, as_view, wg_dtype


class GemmVariant(enum.IntEnum):  # gemm.rs:26-35
    Gemm = 0
    GemmFast = 1
    GemmTr = 2
    GemmTrFast = 3


class GemvVariant(enum.IntEnum):  # gemv.rs:25-34
    Gemv = 0
    GemvFast = 1
    GemvTr = 2
    GemvTrFast = 3


class ReduceOp(enum.IntEnum):  # reduce.rs:13-27
    Min = 0
    Max = 1
    Sum = 2
    Prod = 3
    SqNorm = 4


class OpAssignVariant(enum.IntEnum):  # op_assign.rs:12-26
    Add = 0
    Sub = 1
    Mul = 2
    Div = 3
    Copy = 4


def _common_dtype(*views: GpuTensorView):
    dt = views[0]._tensor.dtype
    for v in views[1:]:
        if v._tensor.dtype is not dt and v._tensor.dtype != dt:
            raise TypeError(f"operands must share one element type, got {[str(x.dtype) for x in views]}")
    return wg_dtype(dt)


def row_major_shader_defs() -> dict:
    """linalg/shape.rs:11-15: the shader definitions that switch `Shape` to row-major.  Pass them to `Gemm.from_device` /
    `Gemv.from_device` (the reference passes them to the shader composer) to get operators whose matrix views are row-major."""
    return {"ROW_MAJOR": True}


def _is_row_major(shader_defs) -> bool:
    return bool(shader_defs) and bool(shader_defs.get("ROW_MAJOR", False))


class Gemm:
    """gemm.rs:9-21.  The four pipelines of the reference are one ahead-of-time compiled MFMA kernel family here, so
    construction is free (no shader compilation)."""

    def __init__(self, device=None, shader_defs=None):
        self.device = device
        self.row_major = _is_row_major(shader_defs)

    @staticmethod
    def from_device(device, shader_defs=None) -> "Gemm":
        return Gemm(device, shader_defs)

    def dispatch(self, device, shapes: ViewShapeBuffers, pass_: ComputePass, out, m1, m2) -> None:
        self.dispatch_generic(device, shapes, pass_, out, m1, m2, GemmVariant.Gemm)

    def dispatch_tr(self, device, shapes: ViewShapeBuffers, pass_: ComputePass, out, m1, m2) -> None:
        self.dispatch_generic(device, shapes, pass_, out, m1, m2, GemmVariant.GemmTr)

    def dispatch_generic(self, device, shapes: ViewShapeBuffers, pass_: ComputePass, out, m1, m2, variant: GemmVariant) -> None:
        out, m1, m2 = as_view(out, 3), as_view(m1, 3), as_view(m2, 3)
        dt = _common_dtype(out, m1, m2)
        fn = lib.wg_gemm_rm if self.row_major else lib.wg_gemm
        check(fn(pass_._ctx.handle, int(variant), dt,
                 out.buffer()._h, out.shape().to_c(), m1.buffer()._h, m1.shape().to_c(),
                 m2.buffer()._h, m2.shape().to_c()))


    def dispatch_ex(self, device, shapes: ViewShapeBuffers, pass_: ComputePass, alpha: float, beta: float, out, m1, m2,
                    variant: GemmVariant = GemmVariant.Gemm) -> None:
        """Extension: out = alpha * op(m1) * m2 + beta * out.  (1, 0) is bit-identical to dispatch_generic."""
        out, m1, m2 = as_view(out, 3), as_view(m1, 3), as_view(m2, 3)
        dt = _common_dtype(out, m1, m2)
        check(lib.wg_gemm_ex(pass_._ctx.handle, int(variant), dt, float(alpha), float(beta),
                             out.buffer()._h, out.shape().to_c(), m1.buffer()._h, m1.shape().to_c(),
                             m2.buffer()._h, m2.shape().to_c()))


class Gemv:
    """gemv.rs:9-21."""

    def __init__(self, device=None, shader_defs=None):
        self.device = device
        self.row_major = _is_row_major(shader_defs)

    @staticmethod
    def from_device(device, shader_defs=None) -> "Gemv":
        return Gemv(device, shader_defs)

    def dispatch(self, device, shapes: ViewShapeBuffers, pass_: ComputePass, out, m, v) -> None:
        self.dispatch_generic(device, shapes, pass_, out, m, v, GemvVariant.Gemv)

    def dispatch_tr(self, device, shapes: ViewShapeBuffers, pass_: ComputePass, out, m, v) -> None:
        self.dispatch_generic(device, shapes, pass_, out, m, v, GemvVariant.GemvTr)

    def dispatch_generic(self, device, shapes: ViewShapeBuffers, pass_: ComputePass, out, m, v, variant: GemvVariant) -> None:
        out, m, v = as_view(out, 3), as_view(m, 3), as_view(v, 3)
        dt = _common_dtype(out, m, v)
        fn = lib.wg_gemv_rm if self.row_major else lib.wg_gemv
        check(fn(pass_._ctx.handle, int(variant), dt,
                 out.buffer()._h, out.shape().to_c(), m.buffer()._h, m.shape().to_c(),
                 v.buffer()._h, v.shape().to_c()))


def gemv_reduce(pass_: ComputePass, op: "ReduceOp", result: GpuTensor, m, v, variant: "GemvVariant" = None) -> None:
    """Extension: result = reduce(op, op(m) v) in one call (scratch vector owned by the context); bit-identical
    to Gemv.dispatch followed by Reduce.dispatch."""
    m, v = as_view(m, 3), as_view(v, 3)
    variant = GemvVariant.Gemv if variant is None else variant
    check(lib.wg_gemv_reduce(pass_._ctx.handle, int(variant), int(op), _common_dtype(m, v), result._h,
                             m.buffer()._h, m.shape().to_c(), v.buffer()._h, v.shape().to_c()))


class Reduce:
    """reduce.rs:62-113: `Reduce::new(device, op)` then `dispatch(device, shapes, pass, value, result)`."""

    def __init__(self, device, op: ReduceOp):
        self.device = device
        self.op = ReduceOp(op)

    @staticmethod
    def new(device, op: ReduceOp) -> "Reduce":
        return Reduce(device, op)

    def dispatch(self, device, shapes: ViewShapeBuffers, pass_: ComputePass, value, result: GpuTensor) -> None:
        value = as_view(value, 1)
        check(lib.wg_reduce(pass_._ctx.handle, int(self.op), wg_dtype(value.dtype), value.buffer()._h, value.shape().to_c(),
                            result._h))

    def dispatch_fast(self, device, shapes: ViewShapeBuffers, pass_: ComputePass, value, result: GpuTensor) -> None:
        """Extension: two-pass multi-workgroup reduce of one long vector at HBM speed. Min/Max: same bits as
        `dispatch`; Sum/Prod/SqNorm: re-associated (deterministic), within n * 2^-24 * sum|x| of the reference order."""
        value = as_view(value, 1)
        check(lib.wg_reduce_fast(pass_._ctx.handle, int(self.op), wg_dtype(value.dtype), value.buffer()._h, value.shape().to_c(),
                                 result._h))

    def dispatch_batched(self, device, shapes: ViewShapeBuffers, pass_: ComputePass, values, results: GpuTensor) -> None:
        """Extension (one launch for every column of a matrix/cube view): results[c + t*ncols], each equal to what
        `dispatch` gives for that column."""
        values = as_view(values, 3)
        check(lib.wg_reduce_batched(pass_._ctx.handle, int(self.op), wg_dtype(values.dtype), values.buffer()._h,
                                    values.shape().to_c(), results._h))

    def eval_cpu(self, val: np.ndarray) -> np.float32:
        """reduce.rs:116-124 (`#[doc(hidden)]` test helper): what nalgebra computes, in NumPy."""
        val = np.asarray(val, np.float32)
        return {ReduceOp.Min: val.min, ReduceOp.Max: val.max, ReduceOp.Prod: val.prod, ReduceOp.Sum: val.sum,
                ReduceOp.SqNorm: lambda: (val * val).sum()}[self.op]()


class OpAssign:
    """op_assign.rs:43-95: `OpAssign::new(device, op)` then `dispatch(device, shapes, pass, in_out_a, in_b)`."""

    def __init__(self, device, op: OpAssignVariant):
        self.device = device
        self.op = OpAssignVariant(op)

    @staticmethod
    def new(device, op: OpAssignVariant) -> "OpAssign":
        return OpAssign(device, op)

    def dispatch(self, device, shapes: ViewShapeBuffers, pass_: ComputePass, in_out_a, in_b) -> None:
        a, b = as_view(in_out_a, 1), as_view(in_b, 1)
        dt = _common_dtype(a, b)
        check(lib.wg_op_assign(pass_._ctx.handle, int(self.op), dt, a.buffer()._h, a.shape().to_c(), b.buffer()._h,
                               b.shape().to_c()))


class Axpy:
    """Extension: the BLAS axpy the north-star names; the reference only has OpAssign (no scalar).
    `dispatch(device, shapes, pass, alpha, in_out_y, in_x)`: y[i] = fma(alpha, x[i], y[i]).  alpha = +1 / -1 give the bits of
    OpAssignVariant.Add / Sub."""

    def __init__(self, device=None):
        self.device = device

    @staticmethod
    def from_device(device) -> "Axpy":
        return Axpy(device)

    def dispatch(self, device, shapes: ViewShapeBuffers, pass_: ComputePass, alpha: float, in_out_y, in_x) -> None:
        y, x = as_view(in_out_y, 1), as_view(in_x, 1)
        dt = _common_dtype(y, x)
        check(lib.wg_axpy(pass_._ctx.handle, float(alpha), dt, y.buffer()._h, y.shape().to_c(), x.buffer()._h, x.shape().to_c()))


class CopyView:
    """Extension: `dispatch(device, shapes, pass, dst, src)`: the dst view = the src view where it has elements, 0 elsewhere; any offset,
    stride and length on either side (`wg_copy_view`). The aligned copy of an odd view, made once instead of inside every product."""

    def __init__(self, device=None):
        self.device = device

    @staticmethod
    def from_device(device) -> "CopyView":
        return CopyView(device)

    def dispatch(self, device, shapes: ViewShapeBuffers, pass_: ComputePass, dst, src) -> None:
        d, s_ = as_view(dst, 3), as_view(src, 3)
        dt = _common_dtype(d, s_)
        check(lib.wg_copy_view(pass_._ctx.handle, dt, d.buffer()._h, d.shape().to_c(), s_.buffer()._h, s_.shape().to_c()))
