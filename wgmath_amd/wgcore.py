"""Host-side mirror of the wgcore surface the dense path uses, over the C ABI (include/wgebra_hip.h).

Same names, argument meaning and error behaviour as the reference so that tests read like the reference's own:
  GpuInstance                         crates/wgcore/src/gpu.rs:7-78
  TensorBuilder / GpuTensor / views   crates/wgcore/src/tensor.rs:41-57,65-187,192-420,445-706
  ViewShape / ViewShapeBuffers        crates/wgcore/src/shapes.rs:9-39,45-117
  CommandEncoder.compute_pass         crates/wgcore/src/kernel.rs:15-27
  GpuTimestamps                       crates/wgcore/src/timestamps.rs
What changes underneath: a buffer is HBM (or pinned host memory for MAP_READ staging tensors), a device+queue is one
in-order HIP stream, a finished encoder can be a replayable hipGraph.  Python is only the test/bench harness language;
every operation below is one call into libwgebra_hip.so.
"""
from __future__ import annotations

import ctypes
import enum
from dataclasses import dataclass, replace
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import ViewShapeC, check, lib

_DT = {np.dtype(np.float32): _lib.WG_F32, np.dtype(np.float16): _lib.WG_F16}


def wg_dtype(dtype) -> int:
    try:
        return _DT[np.dtype(dtype)]
    except KeyError:
        raise TypeError(f"element type {dtype} is not supported by the dense kernels (f32, f16)") from None


class BufferUsages(enum.IntFlag):
    """wgpu::BufferUsages, same bit values."""
    MAP_READ = 1 << 0
    MAP_WRITE = 1 << 1
    COPY_SRC = 1 << 2
    COPY_DST = 1 << 3
    INDEX = 1 << 4
    VERTEX = 1 << 5
    UNIFORM = 1 << 6
    STORAGE = 1 << 7
    INDIRECT = 1 << 8
    QUERY_RESOLVE = 1 << 9


# ---------------------------------------------------------------------------------------------------
# shapes.rs
# ---------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ViewShape:
    """shapes.rs:9-21.  Units: elements."""
    size: tuple  # (rows, cols, mats)
    stride: int
    stride_mat: int
    offset: int

    def to_c(self) -> ViewShapeC:
        c = self.__dict__.get("_c")  # built once per shape object (a dispatch costs microseconds: keep the host side out of it)
        if c is None:
            c = ViewShapeC((ctypes.c_uint32 * 3)(*self.size), self.stride, self.stride_mat, self.offset)
            object.__setattr__(self, "_c", c)
        return c

    def f32_to_vec4(self, column_major: bool = True) -> "ViewShape":
        """shapes.rs:25-38 (floor division, unlike the WGSL twin's ceil)."""
        size = (self.size[0] // 4, self.size[1], self.size[2]) if column_major else (self.size[0], self.size[1] // 4, self.size[2])
        return ViewShape(size, self.stride // 4, self.stride_mat // 4, self.offset // 4)


class ViewShapeBuffers:
    """shapes.rs:45-117.  The reference caches one uniform buffer per distinct ViewShape because WebGPU lacks
    push constants (comment at shapes.rs:43-44).  HIP kernels take the shape by value, so this object holds nothing;
    it exists so that `op.dispatch(device, &shapes, pass, ...)` keeps its signature."""

    def __init__(self):
        self._seen: set = set()

    @staticmethod
    def new() -> "ViewShapeBuffers":
        return ViewShapeBuffers()

    def contains(self, shape: ViewShape) -> bool:
        return shape in self._seen

    def get(self, device, shape: ViewShape) -> ViewShape:
        self._seen.add(shape)
        return shape

    def put_tmp(self, device, queue, shape: ViewShape) -> None:
        self._seen.add(shape)

    def clear_tmp(self) -> None:
        pass


# ---------------------------------------------------------------------------------------------------
# gpu.rs
# ---------------------------------------------------------------------------------------------------
class _Ctx:
    """Owner of the wg_ctx handle."""

    def __init__(self, handle: int, device_index: int):
        self.handle = ctypes.c_void_p(handle)
        self.device_index = device_index
        self.recording = False

    def close(self):
        if self.handle:
            lib.wg_ctx_destroy(self.handle)
            self.handle = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CommandBuffer:
    """wgpu::CommandBuffer.  Eager encoders produce an empty one; recording encoders own a replayable graph."""

    def __init__(self, ctx: _Ctx, handle: Optional[int]):
        self._ctx = ctx
        self._h = ctypes.c_void_p(handle) if handle else None

    def __del__(self):
        if getattr(self, "_h", None):
            if self._ctx.handle:  # the context may already be gone at interpreter shutdown
                lib.wg_cmdbuf_destroy(self._h)
            self._h = None


class ComputePass:
    """wgpu::ComputePass.  Dispatches made through it are enqueued in order on the context's stream."""

    def __init__(self, encoder: "CommandEncoder", label: str, timestamps: Optional["GpuTimestamps"]):
        self.encoder = encoder
        self.label = label
        self._ts = timestamps
        self._ended = False
        if timestamps is not None:
            if hasattr(timestamps, "beginning_of_pass_write_index"):  # ComputePassTimestampWrites: the slots were reserved by next_compute_pass_timestamp_writes
                check(lib.wg_timestamps_write_at(encoder._ctx.handle, timestamps.query_set._h, timestamps.beginning_of_pass_write_index))
            else:
                timestamps._write(encoder._ctx)  # beginning_of_pass_write_index

    @property
    def _ctx(self) -> _Ctx:
        return self.encoder._ctx

    def end(self):
        """Drop for ComputePass (the reference writes `drop(pass)`)."""
        if not self._ended:
            self._ended = True
            if self._ts is not None:
                if hasattr(self._ts, "end_of_pass_write_index"):
                    check(lib.wg_timestamps_write_at(self.encoder._ctx.handle, self._ts.query_set._h, self._ts.end_of_pass_write_index))
                else:
                    self._ts._write(self.encoder._ctx)  # end_of_pass_write_index

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.end()

    def __del__(self):
        try:
            self.end()
        except Exception:
            pass


class CommandEncoder:
    """wgpu::CommandEncoder + wgcore's CommandEncoderExt (kernel.rs:7-27).

    record=False (default): work is enqueued as it is encoded (stream order == encode order, so what the host can
    observe after submit + read is identical to the reference); finish() returns an empty command buffer.
    record=True: encode into a hipGraph; finish() returns a CommandBuffer that Queue.submit can replay many times."""

    def __init__(self, ctx: _Ctx, record: bool = False):
        self._ctx = ctx
        self._record = record
        self._finished = False
        if record:
            check(lib.wg_encoder_begin(ctx.handle))
            ctx.recording = True

    def compute_pass(self, label: str, timestamps: Optional["GpuTimestamps"] = None) -> ComputePass:
        if self._record and timestamps is not None:
            raise ValueError("timestamps cannot be captured into a recorded command buffer")
        return ComputePass(self, label, timestamps)

    def begin_compute_pass(self, label: str = "") -> ComputePass:
        return ComputePass(self, label, None)

    def finish(self) -> CommandBuffer:
        if self._finished:
            raise RuntimeError("CommandEncoder.finish called twice")
        self._finished = True
        if self._record:
            h = ctypes.c_void_p()
            self._ctx.recording = False
            check(lib.wg_encoder_finish(self._ctx.handle, ctypes.byref(h)))
            return CommandBuffer(self._ctx, h.value)
        return CommandBuffer(self._ctx, None)


class Device:
    """wgpu::Device stand-in (what `gpu.device()` returns)."""

    def __init__(self, ctx: _Ctx):
        self._ctx = ctx

    def create_command_encoder(self, desc=None, record: bool = False) -> CommandEncoder:
        return CommandEncoder(self._ctx, record=record)

    def poll_wait(self) -> None:
        """device.poll(PollType::wait())."""
        check(lib.wg_ctx_sync(self._ctx.handle))

    def reserve_workspace(self, nbytes: int) -> None:
        check(lib.wg_ctx_reserve_workspace(self._ctx.handle, nbytes))


class Queue:
    """wgpu::Queue stand-in."""

    def __init__(self, ctx: _Ctx):
        self._ctx = ctx

    def submit(self, command_buffers) -> None:
        if isinstance(command_buffers, CommandBuffer):
            command_buffers = [command_buffers]
        for cb in command_buffers or []:
            if cb is not None and cb._h:
                check(lib.wg_queue_submit(self._ctx.handle, cb._h))

    def get_timestamp_period(self) -> float:
        """wgpu::Queue::get_timestamp_period: nanoseconds per tick of the raw values `GpuTimestamps.wait_for_results` returns (1: they are nanoseconds)."""
        return 1.0

    def write_buffer(self, tensor: "GpuTensor", offset_bytes: int, data) -> None:
        arr = np.ascontiguousarray(data)
        check(lib.wg_buf_write(self._ctx.handle, tensor._h, offset_bytes, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))


class ClockProbe:
    """wg_debug_clock_begin / _end around whatever is enqueued in between on the instance's stream (see GpuInstance.clock_probe)."""

    def __init__(self, gpu):
        self._gpu = gpu
        self._open = lib.wg_debug_clock_begin(gpu._ctx.handle) == _lib.WG_OK  # a diagnostic: never takes the measured run down with it

    def end(self):
        if not self._open:
            return None
        self._open = False
        mean, lo, hi, secs = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        if lib.wg_debug_clock_end(self._gpu._ctx.handle, ctypes.byref(mean), ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(secs)) != _lib.WG_OK:
            return None
        return {"mean": mean.value, "min": lo.value, "max": hi.value, "seconds": secs.value}


class GpuInstance:
    """gpu.rs:7-78.  `GpuInstance.new()` picks device 0; one instance == one GPU == one in-order stream."""

    def __init__(self, ctx: _Ctx):
        self._ctx = ctx
        self._device = Device(ctx)
        self._queue = Queue(ctx)

    @staticmethod
    def new(device_index: int = 0, stream: Optional[int] = None, cu_count: Optional[int] = None, one_xcd: bool = False) -> "GpuInstance":
        """`cu_count`: create the context on a CU-masked stream that may use only that many compute units (multi-GPU runs leave a
        few CUs to the collective library's copy kernels); `one_xcd`: take the missing CUs from one XCD (wg_ctx_create_with_cu_count_one_xcd)."""
        _lib.assert_single_hip_runtime()
        h = ctypes.c_void_p()
        if cu_count is not None:
            create = lib.wg_ctx_create_with_cu_count_one_xcd if one_xcd else lib.wg_ctx_create_with_cu_count
            check(create(device_index, int(cu_count), ctypes.byref(h)))
            inst = GpuInstance(_Ctx(h.value, device_index))
            inst._stream_compute_units = int(cu_count)
            return inst
        elif stream is None:
            check(lib.wg_ctx_create(device_index, ctypes.byref(h)))
        else:
            check(lib.wg_ctx_create_on_stream(device_index, ctypes.c_void_p(stream), ctypes.byref(h)))
        return GpuInstance(_Ctx(h.value, device_index))

    without_gl = new
    with_backends = new

    @staticmethod
    def with_backends(backends=None, device_index: int = 0) -> "GpuInstance":
        """gpu.rs: `GpuInstance::with_backends(backends)` -- there is one backend here (HIP on gfx950); the argument is accepted and ignored."""
        return GpuInstance.new(device_index)

    @staticmethod
    def without_gl(device_index: int = 0) -> "GpuInstance":
        """gpu.rs: `GpuInstance::without_gl()`."""
        return GpuInstance.new(device_index)

    def device_arc(self) -> "Device":
        """gpu.rs: the device behind an `Arc` -- Python references are that already."""
        return self.device()

    @staticmethod
    def device_count() -> int:
        return lib.wg_device_count()

    def adapter(self) -> dict:
        name = ctypes.create_string_buffer(256)
        cus, mhz, hbm = ctypes.c_int(), ctypes.c_int(), ctypes.c_uint64()
        check(lib.wg_ctx_device_info(self._ctx.handle, name, ctypes.byref(cus), ctypes.byref(mhz), ctypes.byref(hbm)))
        info = {"name": name.value.decode(), "compute_units": cus.value, "clock_mhz": mhz.value, "hbm_bytes": hbm.value}
        if getattr(self, "_stream_compute_units", None):  # CU-masked stream (GpuInstance.new(cu_count=...))
            info["stream_compute_units"] = int(self._stream_compute_units)
        return info

    def mem_info(self) -> tuple:
        """(free, total) bytes of device memory right now."""
        free, total = ctypes.c_uint64(), ctypes.c_uint64()
        check(lib.wg_ctx_mem_info(self._ctx.handle, ctypes.byref(free), ctypes.byref(total)))
        return free.value, total.value

    def device(self) -> Device:
        return self._device

    device_arc = device

    def queue(self) -> Queue:
        return self._queue

    def stream(self) -> int:
        return lib.wg_ctx_stream(self._ctx.handle) or 0

    def sync(self) -> None:
        check(lib.wg_ctx_sync(self._ctx.handle))

    # -- diagnostics (wg_debug_*): what the chip did while a kernel ran; the role GpuTimestamps (timestamps.rs:226-230) plays for time ----------
    def clock_probe(self) -> "ClockProbe":
        """Stamps the context's stream now (wg_debug_clock_begin); `.end()` stamps it again, synchronises and returns the mean shader clock of
        the interval in GHz as {"mean", "min", "max"} over the XCDs (+ "seconds"), or None if the probe could not run."""
        return ClockProbe(self)

    def mfma_ceiling(self, min_seconds: float = 0.5) -> dict:
        """v_mfma_f32_16x16x32_f16 alone on random in-register operands for `min_seconds`: {"tflops", "clock_ghz"} (wg_debug_mfma_ceiling)."""
        tf, ghz = ctypes.c_double(), ctypes.c_double()
        check(lib.wg_debug_mfma_ceiling(self._ctx.handle, float(min_seconds), ctypes.byref(tf), ctypes.byref(ghz)))
        return {"tflops": tf.value, "clock_ghz": ghz.value}

    def close(self) -> None:
        self._ctx.close()

    _TUNING = {"f16_tile": _lib.WG_TUNE_F16_TILE, "f16_sched": _lib.WG_TUNE_F16_SCHED, "f32_skinny": _lib.WG_TUNE_F32_SKINNY,
               "f32_panels": _lib.WG_TUNE_F32_PANELS, "f16_balance": _lib.WG_TUNE_F16_BALANCE, "f32_mid": _lib.WG_TUNE_F32_MID, "f32_mid_split": _lib.WG_TUNE_F32_MID_SPLIT, "gemvt_lds": _lib.WG_TUNE_GEMVT_LDS, "f16_cont": _lib.WG_TUNE_F16_CONT, "rm_tr_native": _lib.WG_TUNE_RM_TR_NATIVE}

    def set_tuning(self, knob: str, value: int) -> int:
        """wg_ctx_set_tuning: force a kernel-family choice for later calls on this context (tests / experiments); returns the old value."""
        old = self.get_tuning(knob)
        check(lib.wg_ctx_set_tuning(self._ctx.handle, self._TUNING[knob], int(value)))
        return old

    def f16_balance_info(self) -> dict:
        """wg_ctx_f16_balance_info: measured per-slot rates, snapshots used, launches that ran with calibrated shares."""
        rel, valid, upd, n = (ctypes.c_double * 8)(), ctypes.c_int(), ctypes.c_uint32(), ctypes.c_uint32()
        check(lib.wg_ctx_f16_balance_info(self._ctx.handle, rel, ctypes.byref(valid), ctypes.byref(upd), ctypes.byref(n)))
        return {"rel": [round(float(v), 4) for v in rel], "valid": bool(valid.value), "updates": upd.value, "balanced_launches": n.value}

    def get_tuning(self, knob: str) -> int:
        v = ctypes.c_int()
        check(lib.wg_ctx_get_tuning(self._ctx.handle, self._TUNING[knob], ctypes.byref(v)))
        return v.value


# ---------------------------------------------------------------------------------------------------
# tensor.rs
# ---------------------------------------------------------------------------------------------------
def _prod(shape: Sequence[int]) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n


def _flat_col_major(data, dtype) -> np.ndarray:
    """nalgebra's `as_slice()`: column-major flattening.  1-D input is taken as already flat."""
    arr = np.asarray(data, dtype=dtype)
    return np.ascontiguousarray(arr.reshape(-1, order="F") if arr.ndim > 1 else arr)


class TensorBuilder:
    """tensor.rs:65-187."""

    def __init__(self, shape: Sequence[int], usage: BufferUsages):
        self.shape = tuple(int(s) for s in shape)
        self.usage = BufferUsages(usage)
        self._label = None
        for s in self.shape:
            if not 0 <= s < 2 ** 32:
                raise ValueError("tensor dimensions are u32")

    @staticmethod
    def scalar(usage) -> "TensorBuilder":
        return TensorBuilder((), usage)

    @staticmethod
    def vector(dim: int, usage) -> "TensorBuilder":
        return TensorBuilder((dim,), usage)

    @staticmethod
    def matrix(nrows: int, ncols: int, usage) -> "TensorBuilder":
        return TensorBuilder((nrows, ncols), usage)

    @staticmethod
    def tensor(shape: Sequence[int], usage) -> "TensorBuilder":
        return TensorBuilder(shape, usage)

    def len(self) -> int:
        return _prod(self.shape)

    def label(self, label: str) -> "TensorBuilder":
        self._label = label
        return self

    def build(self, device: Device, dtype=np.float32) -> "GpuTensor":
        """Uninitialised buffer (tensor.rs:115-129)."""
        dt = np.dtype(dtype)
        h = ctypes.c_void_p()
        check(lib.wg_buf_create(device._ctx.handle, self.len() * dt.itemsize, int(self.usage), ctypes.byref(h)))
        return GpuTensor(device._ctx, h.value, self.shape, dt, self.usage)

    def build_init(self, device: Device, data, dtype=None) -> "GpuTensor":
        """tensor.rs:175-186: asserts data.len() >= len, uploads the first `len` elements synchronously."""
        dt = np.dtype(dtype if dtype is not None else getattr(data, "dtype", np.float32))
        flat = _flat_col_major(data, dt)
        n = self.len()
        assert flat.size >= n, (
            "Incorrect number of elements provided for initializing Tensor."
            f"Expected at least {n}, found {flat.size}")
        flat = flat[:n]
        h = ctypes.c_void_p()
        check(lib.wg_buf_create_init(device._ctx.handle, flat.ctypes.data_as(ctypes.c_void_p), flat.nbytes, int(self.usage),
                                     ctypes.byref(h)))
        return GpuTensor(device._ctx, h.value, self.shape, dt, self.usage)

    def build_bytes(self, device: Device, data: bytes, dtype=np.float32) -> "GpuTensor":
        buf = np.frombuffer(data, dtype=np.uint8)
        h = ctypes.c_void_p()
        check(lib.wg_buf_create_init(device._ctx.handle, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes, int(self.usage),
                                     ctypes.byref(h)))
        return GpuTensor(device._ctx, h.value, self.shape, np.dtype(dtype), self.usage)


    def build_uninit_encased(self, device: Device, item_dtype) -> "GpuTensor":
        """tensor.rs:132-147: an uninitialised buffer of `len` items of a shader struct; `item_dtype` is the numpy structured dtype that carries the
        struct's storage layout (its itemsize is the reference's `T::min_size()`, padding included)."""
        return self.build(device, np.dtype(item_dtype))

    def build_encase(self, device: Device, data) -> "GpuTensor":
        """tensor.rs:164-173: items of a shader struct, uploaded in their storage layout. The reference serialises through `encase`; here the layout is the
        structured dtype of `data` (numpy writes the padding the dtype declares), so the bytes that reach the buffer are `data.tobytes()`."""
        arr = np.ascontiguousarray(data)
        assert arr.size >= self.len(), (
            "Incorrect number of elements provided for initializing Tensor."
            f"Expected at least {self.len()}, found {arr.size}")
        return self.build_bytes(device, arr.reshape(-1)[: self.len()].tobytes(), arr.dtype)


class GpuBuffer:
    """What `GpuTensor.into_inner` hands over (tensor.rs:277-279): the device allocation without a shape. Freed when dropped, like the reference's `wgpu::Buffer`."""

    def __init__(self, ctx: _Ctx, handle, nbytes: int):
        self._ctx, self._h, self.size = ctx, handle, int(nbytes)

    def device_ptr(self) -> int:
        return lib.wg_buf_device_ptr(self._h) or 0

    def destroy(self):
        h = getattr(self, "_h", None)
        if h:
            if self._ctx.handle:
                lib.wg_buf_destroy(h)
            self._h = None

    __del__ = destroy


class GpuTensor:
    """tensor.rs:192-400 (GpuScalar / GpuVector / GpuMatrix / GpuCube are DIM = 0..3 of the same type)."""

    def __init__(self, ctx: _Ctx, handle: int, shape: Sequence[int], dtype: np.dtype, usage=BufferUsages.STORAGE, owned=True):
        self._ctx = ctx
        self._h = ctypes.c_void_p(handle)
        self._shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.usage = usage
        self._keepalive = None

    # -- wrapping memory another runtime owns (e.g. a torch tensor): not in the reference, needed for interop --
    @staticmethod
    def wrap(device: Device, device_ptr: int, shape: Sequence[int], dtype=np.float32, keepalive=None) -> "GpuTensor":
        dt = np.dtype(dtype)
        h = ctypes.c_void_p()
        check(lib.wg_buf_wrap(device._ctx.handle, ctypes.c_void_p(device_ptr), _prod(shape) * dt.itemsize, ctypes.byref(h)))
        t = GpuTensor(device._ctx, h.value, shape, dt)
        t._keepalive = keepalive
        return t

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            if self._ctx.handle:  # a buffer must not outlive its context (its handle points into it)
                lib.wg_buf_destroy(h)
            self._h = None

    def destroy(self):
        self.__del__()

    # introspection (tensor.rs:200-279)
    @property
    def DIM(self) -> int:
        return len(self._shape)

    def is_empty(self) -> bool:
        return self.len() == 0

    def len(self) -> int:
        return _prod(self._shape)

    def bytes_len(self) -> int:
        return self.len() * self.dtype.itemsize

    def bytes_len_encased(self) -> int:
        """tensor.rs:217-222: `T::min_size() * len` -- the item size of the storage layout, which the tensor's (possibly structured) dtype carries."""
        return self.bytes_len()

    def shape(self) -> tuple:
        return self._shape

    def buffer(self) -> "GpuTensor":
        return self

    def device_ptr(self) -> int:
        return lib.wg_buf_device_ptr(self._h) or 0

    def into_inner(self) -> "GpuBuffer":
        """tensor.rs:277-279: gives up the tensor, keeps the allocation. The tensor is unusable afterwards (Rust moves it)."""
        assert self._h, "into_inner: the tensor was already consumed"
        inner = GpuBuffer(self._ctx, self._h, self.bytes_len())
        inner._keepalive = self._keepalive
        self._h = None
        return inner

    # copies (tensor.rs:227-264)
    def copy_from(self, encoder: CommandEncoder, source: "GpuTensor") -> None:
        assert self.len() == source.len()
        check(lib.wg_buf_copy(encoder._ctx.handle, source._h, 0, self._h, 0, self.bytes_len()))

    def copy_from_encased(self, encoder: CommandEncoder, source: "GpuTensor") -> None:
        """tensor.rs:235-241: the same buffer-to-buffer copy, sized by the storage layout."""
        assert self.len() == source.len()
        check(lib.wg_buf_copy(encoder._ctx.handle, source._h, 0, self._h, 0, self.bytes_len_encased()))

    def copy_from_view(self, encoder: CommandEncoder, source) -> None:
        source = as_view(source, max(self.DIM, 1))
        assert source.shape().size[0] == (1 if self.DIM == 0 else self._shape[0])
        check(lib.wg_buf_copy(encoder._ctx.handle, source.buffer()._h, source.shape().offset * self.dtype.itemsize, self._h, 0,
                              self.bytes_len()))

    # views (tensor.rs:282-297,514-541)
    def as_view(self) -> "GpuTensorView":
        return self.as_embedded_view(self.DIM)

    def as_embedded_view(self, dim2: int = 3) -> "GpuTensorView":
        assert dim2 >= self.DIM, "Can only embed into a higher-order tensor view."
        cache = self.__dict__.setdefault("_embedded", {})  # views are immutable values: one per dimension is enough
        v = cache.get(dim2)
        if v is None:
            embedded = [1] * dim2
            embedded[:self.DIM] = self._shape
            v = cache[dim2] = self.reshape(embedded, None, None)
        return v

    def reshape(self, shape: Sequence[int], stride: Optional[int] = None, stride_mat: Optional[int] = None) -> "GpuTensorView":
        shape = [int(s) for s in shape]
        # tensor.rs:520 multiplies in u32
        assert (_prod(shape) & 0xFFFFFFFF) <= (_prod(self._shape) & 0xFFFFFFFF)
        size = [1, 1, 1]
        size[:len(shape)] = shape
        s0 = shape[0] if len(shape) > 0 else 1
        s1 = shape[1] if len(shape) > 1 else 1
        default_stride = s0  # column-major
        vs = ViewShape(tuple(size), default_stride if stride is None else stride,
                       (s0 * s1) if stride_mat is None else stride_mat, 0)
        return GpuTensorView(vs, self, len(shape))

    # readback (tensor.rs:300-384)
    def read(self, device: Device) -> np.ndarray:
        """`staging.read(device).await`: blocks until the stream has drained, returns a flat array (Vec<T>)."""
        out = np.empty(self.len(), self.dtype)
        check(lib.wg_buf_read(device._ctx.handle, self._h, 0, out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
        return out

    def read_to(self, device: Device, out: np.ndarray) -> None:
        assert out.dtype == self.dtype and out.size == self.len() and out.flags.c_contiguous
        check(lib.wg_buf_read(device._ctx.handle, self._h, 0, out.ctypes.data_as(ctypes.c_void_p), out.nbytes))

    def read_bytes(self, device: Device) -> bytes:
        return self.read(device).tobytes()

    def slow_read(self, gpu: GpuInstance) -> np.ndarray:
        """tensor.rs:340-355: staging alloc + copy + submit + read."""
        staging = TensorBuilder.tensor(self._shape, BufferUsages.MAP_READ | BufferUsages.COPY_DST).build(gpu.device(), self.dtype)
        enc = gpu.device().create_command_encoder()
        staging.copy_from(enc, self)
        gpu.queue().submit([enc.finish()])
        return staging.read(gpu.device())

    # GpuMatrix / GpuVector / GpuScalar sugar (tensor.rs:544-706)
    @staticmethod
    def uninit(device: Device, *shape_and_usage, dtype=np.float32) -> "GpuTensor":
        *shape, usage = shape_and_usage
        return TensorBuilder(shape, usage).build(device, dtype)

    @staticmethod
    def init(device: Device, data, usage, dtype=None) -> "GpuTensor":
        arr = np.asarray(data)
        return TensorBuilder(arr.shape, usage).build_init(device, arr, dtype or arr.dtype)

    @staticmethod
    def uninit_encased(device: Device, *shape_and_usage, item_dtype) -> "GpuTensor":
        """tensor.rs:553-558,650-655: `GpuMatrix / GpuVector::uninit_encased` -- items of a shader struct, `item_dtype` = the structured dtype with its storage layout."""
        *shape, usage = shape_and_usage
        return TensorBuilder(shape, usage).build_uninit_encased(device, item_dtype)

    @staticmethod
    def encase(device: Device, data, usage) -> "GpuTensor":
        """tensor.rs:633-639: `GpuVector::encase` -- a vector of shader-struct items uploaded in their storage layout (the structured dtype of `data`)."""
        arr = np.ascontiguousarray(data).reshape(-1)
        return TensorBuilder.vector(arr.size, usage).build_encase(device, arr)

    def _require(self, dim: int, what: str):
        if self.DIM != dim:
            raise TypeError(f"{what} is defined on a DIM={dim} tensor, this one has DIM={self.DIM}")

    def column(self, i: int) -> "GpuTensorView":
        self._require(2, "GpuMatrix::column")
        return GpuTensorView(ViewShape((self._shape[0], 1, 1), 1, 1, self._shape[0] * i), self, 1)

    def slice(self, ij, shape) -> "GpuTensorView":
        """tensor.rs:587-594 computes offset = i + j * nrows with the SLICE's nrows (looks like a reference bug;
        reproduced as is -- pass views built by `columns`/`rows` for the conventional meaning)."""
        self._require(2, "GpuMatrix::slice")
        (i, j), (nrows, ncols) = ij, shape
        return GpuTensorView(ViewShape((nrows, ncols, 1), self._shape[0], self._shape[0] * self._shape[1], i + j * nrows), self, 2)

    def columns(self, first_col: int, ncols: int) -> "GpuTensorView":
        self._require(2, "GpuMatrix::columns")
        nrows = self._shape[0]
        return GpuTensorView(ViewShape((nrows, ncols, 1), nrows, self._shape[0] * self._shape[1], first_col * nrows), self, 2)

    def rows(self, first_row: int, nrows: int) -> "GpuTensorView":
        if self.DIM == 1:  # GpuVector::rows, tensor.rs:669-680
            return GpuTensorView(ViewShape((nrows, 1, 1), self._shape[0], self._shape[0], first_row), self, 1)
        self._require(2, "GpuMatrix::rows")
        return GpuTensorView(ViewShape((nrows, self._shape[1], 1), self._shape[0], self._shape[0] * self._shape[1], first_row), self, 2)


GpuScalar = GpuVector = GpuMatrix = GpuCube = GpuTensor


class GpuTensorView:
    """tensor.rs:415-542: non-owning (ViewShape, &Buffer), Copy."""

    def __init__(self, view_shape: ViewShape, tensor: GpuTensor, dim: int):
        self._view_shape = view_shape
        self._tensor = tensor
        self.DIM = dim

    def shape(self) -> ViewShape:
        return self._view_shape

    def buffer(self) -> GpuTensor:
        return self._tensor

    @property
    def dtype(self):
        return self._tensor.dtype

    # GpuVectorView (tensor.rs:436-462)
    def is_empty(self) -> bool:
        return self.len() == 0

    def len(self) -> int:
        return self._view_shape.size[0]

    def rows(self, first: int, nrows: int) -> "GpuTensorView":
        vs = self._view_shape
        if self.DIM <= 1:
            assert first + nrows <= self.len(), f"Rows slice range out of bounds: {first}..{first + nrows}"
            return GpuTensorView(ViewShape((nrows, 1, 1), vs.stride, vs.stride_mat, vs.offset + first), self._tensor, 1)
        return GpuTensorView(ViewShape((nrows, vs.size[1], 1), vs.stride, vs.stride_mat, vs.offset + first), self._tensor, 2)

    # GpuCubeView::matrix (tensor.rs:466-480) -- stride_mat: 1 as in the reference
    def matrix(self, matrix_id: int) -> "GpuTensorView":
        vs = self._view_shape
        assert matrix_id < vs.size[2]
        return GpuTensorView(ViewShape((vs.size[0], vs.size[1], 1), vs.stride, 1, vs.offset + vs.stride_mat * matrix_id), self._tensor, 2)

    # GpuMatrixView::columns (tensor.rs:484-496)
    def columns(self, first_col: int, ncols: int) -> "GpuTensorView":
        vs = self._view_shape
        return GpuTensorView(ViewShape((vs.size[0], ncols, 1), vs.stride, vs.stride_mat, vs.offset + vs.stride * first_col), self._tensor, 2)

    def embedded(self, dim: int) -> "GpuTensorView":
        return GpuTensorView(self._view_shape, self._tensor, dim)


def as_view(x, dim: int = 3) -> GpuTensorView:
    """`impl Into<GpuTensorView<..., DIM>>`: tensors embed (tensor.rs:403-409), views pass through."""
    if isinstance(x, GpuTensorView):
        return x
    if isinstance(x, GpuTensor):
        return x.as_embedded_view(max(dim, x.DIM))
    raise TypeError(f"expected a GpuTensor or GpuTensorView, got {type(x).__name__}")


# ---------------------------------------------------------------------------------------------------
# timestamps.rs
# ---------------------------------------------------------------------------------------------------
class GpuTimestamps:
    """timestamps.rs:9-248: a pool of timestamp slots written at compute-pass boundaries; here hipEvents on the stream."""

    def __init__(self, device: Device, capacity: int):
        self._ctx = device._ctx
        self.capacity = capacity
        h = ctypes.c_void_p()
        check(lib.wg_timestamps_create(self._ctx.handle, capacity, ctypes.byref(h)))
        self._h = h

    @staticmethod
    def new(device: Device, capacity: int) -> "GpuTimestamps":
        return GpuTimestamps(device, capacity)

    def __del__(self):
        if getattr(self, "_h", None):
            if self._ctx.handle:
                lib.wg_timestamps_destroy(self._h)
            self._h = None

    def len(self) -> int:
        return lib.wg_timestamps_len(self._h)

    def is_empty(self) -> bool:
        return self.len() == 0

    def clear(self) -> None:
        check(lib.wg_timestamps_clear(self._h))

    def _write(self, ctx: _Ctx) -> int:
        idx = ctypes.c_uint32()
        check(lib.wg_timestamps_write(ctx.handle, self._h, ctypes.byref(idx)))
        return idx.value

    def write(self, device: Device) -> int:
        return self._write(device._ctx)

    def query_set(self) -> "GpuTimestamps":
        """timestamps.rs:54-57: the underlying query set -- the events live in this object."""
        return self

    # slot allocation (timestamps.rs:59-94): all or nothing, `None` when the slots do not fit
    def next_query_indices(self, count: int) -> Optional[list]:
        first = ctypes.c_uint32()
        check(lib.wg_timestamps_reserve(self._h, int(count), ctypes.byref(first)))
        return None if first.value == 0xFFFFFFFF else [first.value + i for i in range(int(count))]

    def next_query_index(self) -> Optional[int]:
        ids = self.next_query_indices(1)
        return None if ids is None else ids[0]

    def next_compute_pass_timestamp_writes(self) -> Optional["ComputePassTimestampWrites"]:
        """timestamps.rs:59-71: two slots for the beginning and the end of a compute pass: `encoder.compute_pass(label, writes)` writes them."""
        ids = self.next_query_indices(2)
        return None if ids is None else ComputePassTimestampWrites(self, ids[0], ids[1])

    # explicit writes inside a pass (timestamps.rs:96-115)
    def write_next_timestamp(self, compute_pass: ComputePass) -> Optional[int]:
        idx = self.next_query_index()
        if idx is not None:
            check(lib.wg_timestamps_write_at(compute_pass._ctx.handle, self._h, idx))
        return idx

    def write_timestamp_at(self, compute_pass: ComputePass, query_index: int) -> bool:
        if not 0 <= query_index < self.capacity:
            return False
        check(lib.wg_timestamps_write_at(compute_pass._ctx.handle, self._h, int(query_index)))
        return True

    def resolve(self, encoder: CommandEncoder) -> None:
        """timestamps.rs:119-134: a no-op here (events need no resolve copy)."""

    TIMESTAMP_PERIOD_NS = 1.0  # `Queue::get_timestamp_period`: the raw values below are nanoseconds

    def wait_for_results_ms(self, device: Optional[Device] = None, queue=None) -> list:
        """timestamps.rs:226-230: blocks; times in ms relative to the first written timestamp."""
        n = self.len()
        out = (ctypes.c_double * max(n, 1))()
        check(lib.wg_timestamps_wait_for_results_ms(self._h, out, n))
        return [out[i] for i in range(n)]

    def wait_for_results(self, device: Optional[Device] = None) -> list:
        """timestamps.rs:205-224: the raw integer timestamps (here: nanoseconds since the first one; `timestamps_to_ms(raw, queue.get_timestamp_period())` gives ms)."""
        return [int(round(ms * 1.0e6)) for ms in self.wait_for_results_ms()]

    async def wait_for_results_async(self, device: Optional[Device] = None) -> list:
        """timestamps.rs:146-181 (the stream is drained with a blocking wait underneath: there is no map_async here)."""
        return self.wait_for_results(device)

    async def wait_for_results_ms_async(self, queue=None, device: Optional[Device] = None) -> list:
        """timestamps.rs:188-196."""
        return self.wait_for_results_ms(device, queue)

    @staticmethod
    def timestamps_to_ms(timestamps, timestamp_period: float) -> list:
        """timestamps.rs:235-240: raw * period (ns per tick) / 1e6."""
        return [float(t) * float(timestamp_period) / 1.0e6 for t in timestamps]


class ComputePassTimestampWrites:
    """wgpu::ComputePassTimestampWrites as `GpuTimestamps.next_compute_pass_timestamp_writes` hands it out: the query set and the two reserved slots."""

    def __init__(self, query_set: GpuTimestamps, beginning_of_pass_write_index: int, end_of_pass_write_index: int):
        self.query_set = query_set
        self.beginning_of_pass_write_index = beginning_of_pass_write_index
        self.end_of_pass_write_index = end_of_pass_write_index
