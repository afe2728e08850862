"""
wgsl_oracle.py -- NumPy restatement of the wgebra linalg WGSL kernels + ctypes loader for the C one.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; nothing under wgmath_amd/ does.

Two independent restatements live under oracle/:
  * wgsl_oracle.c   -- literal emulation of every WGSL invocation/workgroup (vec4 buffers, u32 index math).
  * this file       -- element-level restatement: each output element's floating-point expression is
                       written out from the WGSL loop structure, vectorised over the output with NumPy.
They must agree BIT-FOR-BIT on vec4-aligned inputs (tests/test_oracle.py); that cross-check is what
guards against a transcription slip in either one.

PARITY STATUS: "parity unpinned" against a real wgpu/naga EXECUTION for Gemm/Gemv/Reduce (the reference cannot run here
and holds no golden vectors), but pinned bit-for-bit against the reference's own shader TEXT executed by oracle/wgsl_exec.py
(tests/golden/wgsl_exec_*.npz, tests/test_oracle.py::test_restatement_matches_executed_wgsl_*) -- see the header of
wgsl_oracle.c and DESIGN.md section 4.

Reference files restated (relative to /root/reference/crates/wgebra/src/linalg/):
  shape.wgsl:36-47,60-66 ; gemm.wgsl:28-200 ; gemv.wgsl:28-155 ; reduce.wgsl:12-96 ; op_assign.wgsl:14-47
  gemm.rs:75-126 ; gemv.rs:74-136 ; reduce.rs:30-58,100-113 ; op_assign.rs:28-38,79-94

Per-element summation orders (K = contraction length, blocks of 4 consecutive k):
  block(jb)   = ((p0 + p1) + p2) + p3 , p_i = a[4jb+i] * b[4jb+i]   (separate IEEE mul / add)
  naive       : acc = 0; for jb = 0..K/4-1: acc = acc + block(jb)                      gemm.wgsl:94-104
  fast (L=64 gemm, L=32 gemv):
                lane l: s_l = 0; for jb = l, l+L, l+2L, ...: s_l = s_l + block(jb)     gemm.wgsl:40-54
                tree:   for stride = L/2 .. 1: s[i] = s[i] + s[i+stride], i < stride   gemm.wgsl:58-63
                (gemv's tree starts at stride 16 on 32 lanes: gemv.wgsl:55-59)
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# enums -- same order as the Rust enums (gemm.rs:26-35, gemv.rs:25-34, reduce.rs:13-27, op_assign.rs:12-26)
GEMM, GEMM_FAST, GEMM_TR, GEMM_TR_FAST = range(4)
GEMV, GEMV_FAST, GEMV_TR, GEMV_TR_FAST = range(4)
MIN, MAX, SUM, PROD, SQNORM = range(5)
ADD, SUB, MUL, DIV, COPY = range(5)

OK, ERR_DIM, ERR_OOB, ERR_ASSERT, ERR_ARG = 0, -1, -2, -3, -4


class OracleError(Exception):
    def __init__(self, code: int, what: str):
        super().__init__(f"{what}: oracle status {code}")
        self.code = code


@dataclass(frozen=True)
class Shape:
    """shape.wgsl:10-33 / wgcore shapes.rs:9-21 (units: elements)."""
    nrows: int
    ncols: int = 1
    nmats: int = 1
    stride: int | None = None
    stride_mat: int | None = None
    offset: int = 0

    def resolved(self) -> "Shape":
        # default (column-major) view of a dense tensor: tensor.rs:514-541
        stride = self.nrows if self.stride is None else self.stride
        stride_mat = self.nrows * self.ncols if self.stride_mat is None else self.stride_mat
        return Shape(self.nrows, self.ncols, self.nmats, stride, stride_mat, self.offset)

    def astuple(self):
        r = self.resolved()
        return (r.nrows, r.ncols, r.nmats, r.stride, r.stride_mat, r.offset)


# --------------------------------------------------------------------------------------------
#                                   NumPy restatement
# --------------------------------------------------------------------------------------------
def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _require_vec4(*shapes: Shape):
    for s in shapes:
        s = s.resolved()
        if s.nrows % 4 or s.stride % 4 or s.stride_mat % 4 or s.offset % 4:
            raise ValueError(
                "NumPy restatement covers vec4-aligned views only (rows/stride/stride_mat/offset % 4 == 0); "
                "for anything else the WGSL result is defined only by the literal emulation in wgsl_oracle.c")


def view(buf: np.ndarray, s: Shape) -> np.ndarray:
    """Gather the (nrows, ncols, nmats) tensor a Shape addresses: idx = t*stride_mat + offset + i + j*stride
    (shape.wgsl:45-47,60-62)."""
    s = s.resolved()
    i = np.arange(s.nrows, dtype=np.int64)[:, None, None]
    j = np.arange(s.ncols, dtype=np.int64)[None, :, None]
    t = np.arange(s.nmats, dtype=np.int64)[None, None, :]
    idx = t * s.stride_mat + s.offset + i + j * s.stride
    if idx.size and idx.max() >= buf.size:
        raise OracleError(ERR_OOB, "view")
    return buf[idx]


def scatter(buf: np.ndarray, s: Shape, values: np.ndarray) -> None:
    s = s.resolved()
    i = np.arange(s.nrows, dtype=np.int64)[:, None, None]
    j = np.arange(s.ncols, dtype=np.int64)[None, :, None]
    t = np.arange(s.nmats, dtype=np.int64)[None, None, :]
    idx = t * s.stride_mat + s.offset + i + j * s.stride
    if idx.size and idx.max() >= buf.size:
        raise OracleError(ERR_OOB, "scatter")
    buf[idx] = values


def _block(a4: np.ndarray, b4: np.ndarray) -> np.ndarray:
    """((p0+p1)+p2)+p3 for one block of 4 consecutive k.  a4: (M,4) rows x k, b4: (4,N) k x cols."""
    t = a4[:, 0:1] * b4[0:1, :]
    t = t + a4[:, 1:2] * b4[1:2, :]
    t = t + a4[:, 2:3] * b4[2:3, :]
    t = t + a4[:, 3:4] * b4[3:4, :]
    return t


def _contract(a_mk: np.ndarray, b_kn: np.ndarray, lanes: int) -> np.ndarray:
    """out(M,N) = a(M,K) @ b(K,N) in the WGSL order.  lanes == 0: naive sequential over blocks;
    lanes == L: lane l takes blocks l, l+L, ...; then the binary tree over lanes."""
    m, k = a_mk.shape
    n = b_kn.shape[1]
    assert k % 4 == 0
    nblk = k // 4
    if lanes == 0:
        acc = np.zeros((m, n), np.float32)
        for jb in range(nblk):
            acc = acc + _block(a_mk[:, 4 * jb:4 * jb + 4], b_kn[4 * jb:4 * jb + 4, :])
        return acc
    assert nblk % lanes == 0, "the *_fast kernels have no tail guard: K % (4*lanes) must be 0"
    part = np.zeros((lanes, m, n), np.float32)
    for jb in range(nblk):
        l = jb % lanes
        part[l] = part[l] + _block(a_mk[:, 4 * jb:4 * jb + 4], b_kn[4 * jb:4 * jb + 4, :])
    stride = lanes // 2
    while stride >= 1:
        part[:stride] = part[:stride] + part[stride:2 * stride]
        stride //= 2
    return part[0]


def gemm(variant: int, out: np.ndarray, so: Shape, m1: np.ndarray, s1: Shape, m2: np.ndarray, s2: Shape) -> None:
    """Gemm::dispatch_generic, gemm.rs:65-127.  Buffers are flat f32 arrays; `out` is written in place."""
    so, s1, s2 = so.resolved(), s1.resolved(), s2.resolved()
    tr = variant in (GEMM_TR, GEMM_TR_FAST)
    fast = variant in (GEMM_FAST, GEMM_TR_FAST)
    m_rows, m_cols = (s1.ncols, s1.nrows) if tr else (s1.nrows, s1.ncols)
    if m_cols != s2.nrows or m_rows != so.nrows or so.ncols != s2.ncols or so.nmats != s1.nmats or so.nmats != s2.nmats:
        raise OracleError(ERR_DIM, "Gemm: dimension mismatch.")
    if out.size == 0 or m1.size == 0 or m2.size == 0 or so.nrows == 0 or so.nmats == 0:
        return
    _require_vec4(so, s1, s2)
    if so.ncols % 4 or m_cols % 4 or m_rows % 4:
        raise ValueError("vec4-aligned shapes only")
    a = view(m1, s1)
    b = view(m2, s2)
    res = np.empty((so.nrows, so.ncols, so.nmats), np.float32)
    for t in range(so.nmats):
        a_mk = a[:, :, t].T if tr else a[:, :, t]
        res[:, :, t] = _contract(np.ascontiguousarray(a_mk), np.ascontiguousarray(b[:, :, t]), 64 if fast else 0)
    scatter(out, so, res)


def gemv(variant: int, out: np.ndarray, so: Shape, m: np.ndarray, sm: Shape, v: np.ndarray, sv: Shape) -> None:
    """Gemv::dispatch_generic, gemv.rs:64-137.  grid.y = RHS column of v/out, grid.z = matrix (m, v, out alike)."""
    so, sm, sv = so.resolved(), sm.resolved(), sv.resolved()
    tr = variant in (GEMV_TR, GEMV_TR_FAST)
    m_rows, m_cols = (sm.ncols, sm.nrows) if tr else (sm.nrows, sm.ncols)
    if m_cols != sv.nrows or m_rows != so.nrows:
        raise OracleError(ERR_DIM, "Gemv: dimension mismatch.")
    if variant == GEMV_TR_FAST and sm.nrows % 128 != 0:
        variant = GEMV_TR  # gemv.rs:99-104
    fast = variant in (GEMV_FAST, GEMV_TR_FAST)
    if fast and so.nrows % 4 != 0:
        raise OracleError(ERR_ASSERT, "gemv.rs:122 assert_eq!(out_nrows % 4, 0)")
    if out.size == 0 or m.size == 0 or v.size == 0 or so.nrows == 0 or so.ncols == 0 or so.nmats == 0:
        return
    _require_vec4(so, sm, sv)
    if m_cols % 4 or m_rows % 4:
        raise ValueError("vec4-aligned shapes only")
    # the kernel indexes m, v with grid (y, z) taken from OUT's ncols/nmats (gemv.rs:136) -- emulate that.
    sm_g = Shape(sm.nrows, sm.ncols, so.nmats, sm.stride, sm.stride_mat, sm.offset)
    sv_g = Shape(sv.nrows, so.ncols, so.nmats, sv.stride, sv.stride_mat, sv.offset)
    a = view(m, sm_g)
    x = view(v, sv_g)
    res = np.empty((so.nrows, so.ncols, so.nmats), np.float32)
    for t in range(so.nmats):
        a_mk = a[:, :, t].T if tr else a[:, :, t]
        res[:, :, t] = _contract(np.ascontiguousarray(a_mk), np.ascontiguousarray(x[:, :, t]), 32 if fast else 0)
    scatter(out, so, res)


_INIT = {MIN: np.float32(3.4e38), MAX: np.float32(-3.4e38), SUM: np.float32(0), PROD: np.float32(1), SQNORM: np.float32(0)}


def reduce_vec(op: int, x: np.ndarray) -> np.float32:
    """reduce.wgsl:68-87 on a contiguous f32 vector: 128 strided lanes, then the 64..1 tree."""
    x = _f32(x)
    n = x.size
    ws = np.full(128, _INIT[op], np.float32)
    rows = -(-n // 128)
    for r in range(rows):
        seg = x[128 * r:128 * r + 128]
        k = seg.size
        if op == MIN:
            ws[:k] = np.minimum(ws[:k], seg)
        elif op == MAX:
            ws[:k] = np.maximum(ws[:k], seg)
        elif op == SUM:
            ws[:k] = ws[:k] + seg
        elif op == PROD:
            ws[:k] = ws[:k] * seg
        else:
            ws[:k] = ws[:k] + seg * seg
    stride = 64
    while stride >= 1:
        a, b = ws[:stride], ws[stride:2 * stride]
        if op == MIN:
            ws[:stride] = np.minimum(a, b)
        elif op == MAX:
            ws[:stride] = np.maximum(a, b)
        elif op == PROD:
            ws[:stride] = a * b
        else:
            ws[:stride] = a + b
        stride //= 2
    return ws[0]


def reduce(op: int, inp: np.ndarray, shape: Shape) -> np.float32:
    """Reduce::dispatch, reduce.rs:100-113: input read at offset+i, i < nrows (stride ignored)."""
    s = shape.resolved()
    if s.offset + s.nrows > inp.size:
        raise OracleError(ERR_OOB, "reduce")
    with np.errstate(over="ignore", under="ignore"):
        return reduce_vec(op, inp[s.offset:s.offset + s.nrows])


def reduce_batched(op: int, inp: np.ndarray, shape: Shape) -> np.ndarray:
    s = shape.resolved()
    out = np.empty(s.ncols * s.nmats, np.float32)
    for t in range(s.nmats):
        for c in range(s.ncols):
            off = s.offset + c * s.stride + t * s.stride_mat
            out[c + t * s.ncols] = reduce(op, inp, Shape(s.nrows, 1, 1, s.stride, s.stride_mat, off))
    return out


def op_assign(op: int, a: np.ndarray, sa: Shape, b: np.ndarray, sb: Shape) -> None:
    """OpAssign::dispatch, op_assign.rs:71-95 + op_assign.wgsl:40-47 (in place on `a`)."""
    sa, sb = sa.resolved(), sb.resolved()
    if sa.nrows != sb.nrows:
        raise OracleError(ERR_DIM, "Op-assign: dimension mismatch.")
    if a.size == 0 or b.size == 0 or sa.nrows == 0:
        return
    n = sa.nrows
    if sa.offset + n > a.size or sb.offset + n > b.size:
        raise OracleError(ERR_OOB, "op_assign")
    av = a[sa.offset:sa.offset + n]
    bv = b[sb.offset:sb.offset + n].copy()
    with np.errstate(all="ignore"):
        if op == ADD:
            av[:] = av + bv
        elif op == SUB:
            av[:] = av - bv
        elif op == MUL:
            av[:] = av * bv
        elif op == DIV:
            av[:] = av / bv
        else:
            av[:] = bv


# --------------------------------------------------------------------------------------------
#                             ctypes loader for the C restatement
# --------------------------------------------------------------------------------------------
class _CShape(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("nrows", "ncols", "nmats", "stride", "stride_mat", "offset")]


def _cshape(s: Shape) -> _CShape:
    return _CShape(*s.astuple())


def build_c(force: bool = False) -> str:
    so = os.path.join(_HERE, "_build", "libwgsl_oracle.so")
    src = os.path.join(_HERE, "wgsl_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return so


class CLib:
    """The C restatement.  All buffers are flat, C-contiguous float32 NumPy arrays."""

    def __init__(self):
        self.lib = ctypes.CDLL(build_c())
        fp = ctypes.POINTER(ctypes.c_float)
        u64, u32, ci = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
        self.lib.wgo_gemm.argtypes = [ci, fp, u64, _CShape, fp, u64, _CShape, fp, u64, _CShape, u32, u32]
        self.lib.wgo_gemv.argtypes = [ci, fp, u64, _CShape, fp, u64, _CShape, fp, u64, _CShape, u32, u32]
        self.lib.wgo_reduce.argtypes = [ci, fp, u64, _CShape, fp, u64]
        self.lib.wgo_reduce_batched.argtypes = [ci, fp, u64, _CShape, fp, u64]
        self.lib.wgo_op_assign.argtypes = [ci, fp, u64, _CShape, fp, u64, _CShape]
        self.lib.wgo_axpy.argtypes = [ctypes.c_float, fp, u64, _CShape, fp, u64, _CShape]
        self.lib.wgo_axpy.restype = ci
        for f in ("wgo_gemm", "wgo_gemv", "wgo_reduce", "wgo_reduce_batched", "wgo_op_assign", "wgo_num_threads"):
            getattr(self.lib, f).restype = ci
        self.lib.wgo_set_num_threads.argtypes = [ci]
        self.lib.wgo_version.restype = ctypes.c_char_p

    @staticmethod
    def _p(a: np.ndarray):
        assert a.dtype == np.float32 and a.flags.c_contiguous and a.ndim == 1
        return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.c_uint64(a.size)

    @staticmethod
    def _chk(rc: int, what: str):
        if rc != OK:
            raise OracleError(rc, what)

    def num_threads(self) -> int:
        return self.lib.wgo_num_threads()

    def set_num_threads(self, n: int) -> None:
        self.lib.wgo_set_num_threads(n)

    def gemm(self, variant, out, so, m1, s1, m2, s2, wg_begin=0, wg_end=0xFFFFFFFF):
        self._chk(self.lib.wgo_gemm(variant, *self._p(out), _cshape(so), *self._p(m1), _cshape(s1),
                                    *self._p(m2), _cshape(s2), wg_begin, wg_end), "Gemm")

    def gemv(self, variant, out, so, m, sm, v, sv, wg_begin=0, wg_end=0xFFFFFFFF):
        self._chk(self.lib.wgo_gemv(variant, *self._p(out), _cshape(so), *self._p(m), _cshape(sm),
                                    *self._p(v), _cshape(sv), wg_begin, wg_end), "Gemv")

    def reduce(self, op, inp, shape) -> np.float32:
        res = np.zeros(1, np.float32)
        self._chk(self.lib.wgo_reduce(op, *self._p(inp), _cshape(shape), *self._p(res)), "Reduce")
        return res[0]

    def reduce_batched(self, op, inp, shape) -> np.ndarray:
        s = shape.resolved()
        res = np.zeros(s.ncols * s.nmats, np.float32)
        self._chk(self.lib.wgo_reduce_batched(op, *self._p(inp), _cshape(shape), *self._p(res)), "Reduce(batched)")
        return res

    def op_assign(self, op, a, sa, b, sb):
        self._chk(self.lib.wgo_op_assign(op, *self._p(a), _cshape(sa), *self._p(b), _cshape(sb)), "OpAssign")

    def axpy(self, alpha, y, sy, x, sx):
        self._chk(self.lib.wgo_axpy(ctypes.c_float(alpha), *self._p(y), _cshape(sy), *self._p(x), _cshape(sx)), "Axpy")


# --------------------------------------------------------------------------------------------
#                 f64 ground truth + the K-scaled tolerance of DESIGN.md / SURVEY 8(c)
# --------------------------------------------------------------------------------------------
def gemm_f64(a_mk: np.ndarray, b_kn: np.ndarray):
    """Returns (exact-ish f64 product, sum_k |a||b|) for the error bound."""
    a64, b64 = a_mk.astype(np.float64), b_kn.astype(np.float64)
    return a64 @ b64, np.abs(a64) @ np.abs(b64)


def f32_tolerance(k: int, sum_abs: np.ndarray, c: float = 2.0) -> np.ndarray:
    """Realistic gate c*sqrt(K)*2^-24*sum|a||b| (c >= 2 also covers the final rounding); hard bound: hard_bound()."""
    return c * np.sqrt(max(k, 1)) * 2.0 ** -24 * sum_abs + np.finfo(np.float32).tiny


def hard_bound(k: int, sum_abs: np.ndarray) -> np.ndarray:
    return (k + 1) * 2.0 ** -24 * sum_abs + np.finfo(np.float32).tiny
