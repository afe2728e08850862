"""ctypes binding of libwgebra_hip.so (include/wgebra_hip.h).  No fallback: if the HIP library is missing or an
entry point is absent, importing this module raises -- the product path never routes around the GPU kernels."""
from __future__ import annotations

import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
# WGEBRA_HIP_LIB selects another build of the SAME library (kernel A/B experiments); never a different backend.
LIB_PATH = os.environ.get("WGEBRA_HIP_LIB") or os.path.join(_PKG, "libwgebra_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "wgebra_hip.h")

# status codes (wg_status)
WG_OK, WG_ERR_DIM_MISMATCH, WG_ERR_PRECONDITION, WG_ERR_INVALID_ARG, WG_ERR_OUT_OF_BOUNDS, WG_ERR_HIP, \
    WG_ERR_UNSUPPORTED, WG_ERR_NO_DEVICE, WG_ERR_WORKSPACE = range(9)
WG_GATHER_RCCL, WG_GATHER_NONE, WG_GATHER_PEER_STAGED = 0, 2, 3  # (1 was the SDMA rect-copy engine: removed in ABI 3)
WG_COMM_ID_BYTES, WG_IPC_HANDLE_BYTES = 128, 96
ABI_VERSION = 4  # == WGEBRA_HIP_ABI_VERSION (checked when the library is loaded, and against the header by tests/test_abi_and_host.py)
WG_F32, WG_F16 = 0, 1
WG_TUNE_F16_TILE, WG_TUNE_F16_SCHED, WG_TUNE_F32_SKINNY, WG_TUNE_F32_PANELS, WG_TUNE_F16_BALANCE, WG_TUNE_F32_MID, WG_TUNE_F32_MID_SPLIT, WG_TUNE_GEMVT_LDS, WG_TUNE_F16_CONT, WG_TUNE_RM_TR_NATIVE = range(10)


class ViewShapeC(ctypes.Structure):
    """wg_view_shape == wgcore::shapes::ViewShape (#[repr(C)], 24 bytes; shapes.rs:9-21)."""
    _fields_ = [("size", ctypes.c_uint32 * 3), ("stride", ctypes.c_uint32), ("stride_mat", ctypes.c_uint32),
                ("offset", ctypes.c_uint32)]


assert ctypes.sizeof(ViewShapeC) == 24


class WgError(RuntimeError):
    """A non-OK wg_status.  `.status` is the code, the message is wg_last_error_string()."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


class DimensionMismatch(WgError, AssertionError):
    """What the reference raises as a panic from assert_eq!(..., "... dimension mismatch.")."""


class PreconditionFailed(WgError, AssertionError):
    pass


class NoDevice(WgError):
    pass


class WorkspaceMustGrow(WgError):
    """A context scratch region would have to grow inside a recording (WG_ERR_WORKSPACE): run the call once eagerly first."""


def declared_symbols(header_path: str = HEADER_PATH) -> list[str]:
    """Every function the public header declares (used by the ABI-completeness test)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wg_[a-z0-9_]+)\s*\(", text)))


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C wgmath_amd/csrc`. There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    pd = ctypes.POINTER(ctypes.c_double)
    vp, cp, ci, u32, u64, sz = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t
    pvp = ctypes.POINTER(vp)
    S = ViewShapeC
    sig = {
        "wg_abi_version": (ci, []),
        "wg_last_error_string": (cp, []),
        "wg_device_count": (ci, []),
        "wg_ctx_create": (ci, [ci, pvp]),
        "wg_ctx_create_on_stream": (ci, [ci, vp, pvp]),
        "wg_ctx_create_with_cu_count": (ci, [ci, ctypes.c_uint32, ctypes.POINTER(vp)]),
        "wg_ctx_create_with_cu_count_one_xcd": (ci, [ci, ctypes.c_uint32, ctypes.POINTER(vp)]),
        "wg_ctx_destroy": (ci, [vp]),
        "wg_ctx_sync": (ci, [vp]),
        "wg_ctx_device": (ci, [vp]),
        "wg_ctx_stream": (vp, [vp]),
        "wg_ctx_device_info": (ci, [vp, cp, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(u64)]),
        "wg_ctx_mem_info": (ci, [vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]),
        "wg_ctx_reserve_workspace": (ci, [vp, sz]),
        "wg_debug_f16_balance_plan": (ci, [ctypes.POINTER(ctypes.c_double), u32, u32, ci, ctypes.POINTER(u32), u32, ctypes.POINTER(u32), ctypes.POINTER(u32)]),
        "wg_ctx_f16_balance_info": (ci, [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ci), ctypes.POINTER(u32), ctypes.POINTER(u32)]),
        "wg_ctx_set_tuning": (ci, [vp, ci, ci]),
        "wg_ctx_get_tuning": (ci, [vp, ci, ctypes.POINTER(ci)]),
        "wg_debug_spin": (ci, [vp, ctypes.c_uint32, ctypes.c_uint32, vp]),
        "wg_geometry_apply": (ci, [vp, ci, ctypes.c_uint32, vp, vp, ctypes.c_uint32]),
        "wg_buf_create": (ci, [vp, sz, u32, pvp]),
        "wg_buf_create_init": (ci, [vp, vp, sz, u32, pvp]),
        "wg_buf_wrap": (ci, [vp, vp, sz, pvp]),
        "wg_buf_destroy": (ci, [vp]),
        "wg_buf_size": (sz, [vp]),
        "wg_buf_device_ptr": (vp, [vp]),
        "wg_buf_write": (ci, [vp, vp, sz, vp, sz]),
        "wg_buf_read": (ci, [vp, vp, sz, vp, sz]),
        "wg_buf_copy": (ci, [vp, vp, sz, vp, sz, sz]),
        "wg_buf_fill_zero": (ci, [vp, vp]),
        "wg_gemm": (ci, [vp, ci, ci, vp, S, vp, S, vp, S]),
        "wg_gemm_ex": (ci, [vp, ci, ci, ctypes.c_float, ctypes.c_float, vp, S, vp, S, vp, S]),
        "wg_gemv": (ci, [vp, ci, ci, vp, S, vp, S, vp, S]),
        "wg_gemm_rm": (ci, [vp, ci, ci, vp, S, vp, S, vp, S]),
        "wg_gemv_rm": (ci, [vp, ci, ci, vp, S, vp, S, vp, S]),
        "wg_gemv_reduce": (ci, [vp, ci, ci, ci, vp, vp, S, vp, S]),
        "wg_reduce": (ci, [vp, ci, ci, vp, S, vp]),
        "wg_reduce_fast": (ci, [vp, ci, ci, vp, S, vp]),
        "wg_reduce_batched": (ci, [vp, ci, ci, vp, S, vp]),
        "wg_op_assign": (ci, [vp, ci, ci, vp, S, vp, S]),
        "wg_axpy": (ci, [vp, ctypes.c_float, ci, vp, S, vp, S]),
        "wg_copy_view": (ci, [vp, ci, vp, S, vp, S]),
        "wg_debug_clock_begin": (ci, [vp]),
        "wg_debug_clock_end": (ci, [vp, pd, pd, pd, pd]),
        "wg_debug_mfma_ceiling": (ci, [vp, ctypes.c_double, pd, pd]),
        "wg_comm_unique_id": (ci, [vp]),
        "wg_comm_create": (ci, [vp, ci, ci, vp, pvp]),
        "wg_comm_destroy": (ci, [vp]),
        "wg_comm_rank": (ci, [vp]),
        "wg_comm_size": (ci, [vp]),
        "wg_comm_has_collectives": (ci, [vp]),
        "wg_comm_reported_size": (ci, [vp, ctypes.POINTER(ci)]),
        "wg_comm_bytes_sent": (u64, [vp]),
        "wg_all_gather": (ci, [vp, ci, vp, u64, u64]),
        "wg_comm_join": (ci, [vp]),
        "wg_comm_set_pipelined": (ci, [vp, ci]),
        "wg_comm_set_one_launch": (ci, [vp, ci]),
        "wg_comm_flush": (ci, [vp]),
        "wg_comm_barrier": (ci, [vp]),
        "wg_buf_ipc_export": (ci, [vp, vp]),
        "wg_buf_ipc_open": (ci, [vp, vp, pvp]),
        "wg_comm_stage_reserve": (ci, [vp, sz, pvp, pvp]),
        "wg_comm_set_peer_stages": (ci, [vp, pvp, pvp]),
        "wg_cube_to_matrix": (ci, [vp, ci, vp, S, vp, S]),
        "wg_gemm_sharded": (ci, [vp, ci, ci, ci, u32, vp, S, vp, S, vp, S]),
        "wg_comm_set_wait_timing": (ci, [vp, ci]),
        "wg_comm_wait_times": (ci, [vp, ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_float), u32, ctypes.POINTER(u32)]),
        "wg_gemm_sharded_panels": (ci, [vp, ci, ci, ci, ctypes.POINTER(u32), u32, vp, S, vp, S, vp, S]),
        "wg_encoder_begin": (ci, [vp]),
        "wg_encoder_finish": (ci, [vp, pvp]),
        "wg_queue_submit": (ci, [vp, vp]),
        "wg_cmdbuf_destroy": (ci, [vp]),
        "wg_timestamps_create": (ci, [vp, u32, pvp]),
        "wg_timestamps_destroy": (ci, [vp]),
        "wg_timestamps_clear": (ci, [vp]),
        "wg_timestamps_write": (ci, [vp, vp, ctypes.POINTER(u32)]),
        "wg_timestamps_reserve": (ci, [vp, u32, ctypes.POINTER(u32)]),
        "wg_timestamps_write_at": (ci, [vp, vp, u32]),
        "wg_timestamps_len": (u32, [vp]),
        "wg_timestamps_wait_for_results_ms": (ci, [vp, ctypes.POINTER(ctypes.c_double), u32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it: fail loudly
        fn.restype = res
        fn.argtypes = args
    lib._wg_signatures = sig
    # a stale build of the library with every symbol this binding names but other semantics must not pass for the current one
    if lib.wg_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} reports ABI version {lib.wg_abi_version()}, this binding is written for {ABI_VERSION} "
                          f"(include/wgebra_hip.h: WGEBRA_HIP_ABI_VERSION): rebuild with `make -C wgmath_amd/csrc`")
    return lib


def hip_runtime_paths() -> list[str]:
    """Distinct libamdhip64 images mapped into this process."""
    paths = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    paths.add(line.split()[-1])
    except OSError:
        pass
    return sorted(paths)


def assert_single_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64. If this library was loaded BEFORE torch, the
    process ends up with two HIP runtimes and the second one cannot open the GPU ("no HIP device visible"). Import torch
    first whenever both are used in one process (bench.py does for --gpus > 1)."""
    p = hip_runtime_paths()
    if len(p) > 1:
        raise ImportError("two HIP runtimes are loaded in this process: " + ", ".join(p) +
                          ". Import torch BEFORE wgmath_amd so that libwgebra_hip.so binds to torch's bundled runtime.")


lib = _load()

_EXC = {WG_ERR_DIM_MISMATCH: DimensionMismatch, WG_ERR_PRECONDITION: PreconditionFailed, WG_ERR_NO_DEVICE: NoDevice, WG_ERR_WORKSPACE: WorkspaceMustGrow}


def check(status: int) -> None:
    if status != WG_OK:
        msg = lib.wg_last_error_string().decode("utf-8", "replace")
        raise _EXC.get(status, WgError)(status, msg or f"wg_status {status}")
