"""wgmath_amd: MI355X (gfx950) backend behind the wgebra dense operator surface.

The product is libwgebra_hip.so (hand-written HIP kernels + C ABI, sources in csrc/, header in include/wgebra_hip.h).
This package is the thin host-side mirror of the reference's wgcore/wgebra API used by the tests and the bench.
Importing it loads the shared library and fails loudly if it is not built: there is no CPU or PyTorch fallback.
"""
from ._lib import (LIB_PATH, DimensionMismatch, NoDevice, PreconditionFailed, WgError, WorkspaceMustGrow)  # noqa: F401
from .wgcore import (BufferUsages, CommandBuffer, CommandEncoder, ComputePass, Device, GpuCube, GpuInstance,  # noqa: F401
                     GpuBuffer, GpuMatrix, GpuScalar, GpuTensor, GpuTensorView, GpuTimestamps, ComputePassTimestampWrites, GpuVector, Queue, TensorBuilder,
                     ViewShape, ViewShapeBuffers, as_view)
from .wgebra import (Axpy, CopyView, Gemm, GemmVariant, Gemv, GemvVariant, OpAssign, OpAssignVariant, Reduce, ReduceOp,  # noqa: F401
                     gemv_reduce, row_major_shader_defs)
from .sharded import Comm, GatherMode, MShardPlan, ShardedGemm, new_unique_id  # noqa: F401,E402
from . import geometry  # noqa: F401,E402
from .geometry import GeomOp  # noqa: F401,E402

__version__ = "0.1.0"
