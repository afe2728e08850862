"""Batched entry to the geometry header (include/wgebra_geometry.hpp), the HIP counterpart of the per-function WGSL modules under
crates/wgebra/src/geometry/ (inv.rs, cholesky.rs, lu.rs, qr2/3/4.rs, eig2/3/4.rs, svd2/3.rs, rot2.rs, quat.rs, sim2.rs, sim3.rs).
The reference exposes those as shader libraries other shaders import; its tests drive each through a one-item-per-invocation kernel
over a storage array, which is what `apply` launches. Item layouts: include/wgebra_hip.h (wg_geom_op)."""
import enum

from ._lib import check, lib
from .wgcore import GpuInstance, GpuTensor


class GeomOp(enum.IntEnum):
    INV = 0
    CHOLESKY = 1
    LU = 2
    QR = 3
    SYM_EIGEN = 4
    SVD = 5
    ROT2 = 6
    QUAT = 7
    SIM2 = 8
    SIM3 = 9
    QUAT_RAW = 10  # the transform functions one by one on raw coordinates (layouts: csrc/geometry_items.hpp raw_item)
    ROT2_RAW = 11
    SIM2_RAW = 12
    SIM3_RAW = 13
    FROM = 14      # quat::fromScaledAxis + rot2::fromAngle
    UTILS = 15     # wgebra::trig (stable_atan2, stable_tanh) + wgebra::min_max
    ROT2_EXT = 16  # rot2: angle, cancel_y, is_valid, rotate_rows3, rotate_rows4
    EIGVALS2 = 17  # eig2::eigenvalues
    SVD_RECOMPOSE = 18  # svd2 / svd3 ::recompose (dim 2, 3)


_FIXED_IN = {GeomOp.ROT2: 4, GeomOp.QUAT: 9, GeomOp.SIM2: 10, GeomOp.SIM3: 17, GeomOp.QUAT_RAW: 11, GeomOp.ROT2_RAW: 6, GeomOp.SIM2_RAW: 12,
             GeomOp.SIM3_RAW: 19, GeomOp.FROM: 4, GeomOp.UTILS: 19, GeomOp.ROT2_EXT: 31, GeomOp.EIGVALS2: 4}
_FIXED_OUT = {GeomOp.ROT2: 11, GeomOp.QUAT: 19, GeomOp.SIM2: 14, GeomOp.SIM3: 25, GeomOp.QUAT_RAW: 27, GeomOp.ROT2_RAW: 12, GeomOp.SIM2_RAW: 18,
              GeomOp.SIM3_RAW: 28, GeomOp.FROM: 6, GeomOp.UTILS: 11, GeomOp.ROT2_EXT: 29, GeomOp.EIGVALS2: 2}


def in_floats(op: GeomOp, dim: int = 0) -> int:
    if GeomOp(op) == GeomOp.SVD_RECOMPOSE:
        return 2 * dim * dim + dim
    return _FIXED_IN.get(GeomOp(op), dim * dim)


def out_floats(op: GeomOp, dim: int = 0) -> int:
    op = GeomOp(op)
    if op in _FIXED_OUT:
        return _FIXED_OUT[op]
    n = dim
    return {GeomOp.INV: n * n, GeomOp.CHOLESKY: n * n, GeomOp.LU: n * n + 2 * n + 1, GeomOp.QR: 2 * n * n,
            GeomOp.SYM_EIGEN: n * n + n, GeomOp.SVD: 2 * n * n + n, GeomOp.SVD_RECOMPOSE: n * n}[op]


def apply(gpu: GpuInstance, op: GeomOp, dim: int, items: GpuTensor, out: GpuTensor, count: int) -> None:
    """One invocation per item on the instance's stream (asynchronous, like a dispatch); raises WgError on bad arguments."""
    check(lib.wg_geometry_apply(gpu._ctx.handle, int(op), int(dim), items._h, out._h, int(count)))
