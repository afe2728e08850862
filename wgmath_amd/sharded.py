"""M-sharded GEMM across the GPUs of one node: one process per GPU, RCCL all-gather over xGMI.

Two layers:
  * `Comm` / `sharded_gemm` -- the product path: thin ctypes callers of the C ABI's multi-GPU section (include/wgebra_hip.h:
    wg_comm_*, wg_gemm_sharded). RCCL is driven by the library itself (no torch needed); every rank ends with a plain M x N
    column-major GpuMatrix. Exchange engines: RCCL all-gather of a staging cube + relayout, or contiguous SDMA peer copies of the
    cube's slots (one engine per link) + relayout (csrc/comm.hip).
  * `MShardPlan` / `ShardedGemm` -- the planner and the cube-layout driver with injected GEMM / collective (pure host arithmetic,
    runs under gloo in the CPU tests): the gathered result stays a GpuCube per panel, no relayout.

The reference has no multi-device path (one wgpu::Device + one Queue, crates/wgcore/src/gpu.rs:7-12); this is the
north-star's addition, expressed in the reference's own tensor model:

  * rank g owns A_g = A[g*Mg:(g+1)*Mg, :] as its OWN contiguous column-major tensor (Mg = M/P, a multiple of 4 so the vec4
    precondition holds), B (K x N) is replicated, C_g = A_g * B needs no communication;
  * row blocks of a column-major C are not contiguous, so the gathered result is not an M x N matrix but, per N-panel, a
    GpuCube [Mg, np, P] (stride = Mg, stride_mat = np*Mg): matrix g of the cube is rank g's row block of that panel --
    exactly what `GpuCubeView::matrix(g)` addresses (tensor.rs:466-480). The gathered buffer is laid out
    [panel][rank][np*Mg] so that every panel's all-gather writes one contiguous range (in-place: each rank's GEMM writes
    its own slot of the panel, then the collective fills the others).
  * N is split into panels so that the all-gather of panel i (on RCCL's stream) overlaps the GEMM of panel i+1: xGMI is
    point-to-point (7 links x ~153 GB/s per GPU; only P-1 of them reach participating peers), so an un-overlapped gather
    would cost ~40 % of the 4-GPU compute time at 32768^3.

`MShardPlan` is pure host arithmetic (tested on CPU); `ShardedGemm` drives one rank. The local GEMM and the collective are
injected so that the same driver runs under gloo in the CPU tests and on the HIP kernels + RCCL in bench.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import ctypes

from .wgcore import ViewShape


@dataclass(frozen=True)
class MShardPlan:
    M: int
    N: int
    K: int
    world: int
    npanels: int = 1
    # optional ragged panels: column count of every panel (multiples of 4, summing to N; overrides the uniform N / npanels).
    # bench.py sizes panels to a whole number of rounds of the CUs the compute stream may use, plus one remainder panel.
    panel_cols: Optional[Tuple[int, ...]] = None

    def __post_init__(self):
        if self.panel_cols is not None:
            object.__setattr__(self, "panel_cols", tuple(int(c) for c in self.panel_cols))
            object.__setattr__(self, "npanels", len(self.panel_cols))
            if not self.panel_cols or sum(self.panel_cols) != self.N or any(c <= 0 or c % 4 for c in self.panel_cols):
                raise ValueError(f"panel_cols={self.panel_cols} must be positive multiples of 4 summing to N={self.N}")
        if self.world < 1 or self.npanels < 1:
            raise ValueError("world and npanels must be >= 1")
        if self.M % (4 * self.world):
            raise ValueError(f"M={self.M} must split into {self.world} row blocks that are multiples of 4 (vec4 views)")
        if self.panel_cols is None and self.N % (4 * self.npanels):
            raise ValueError(f"N={self.N} must split into {self.npanels} column panels that are multiples of 4")
        if self.K % 4:
            raise ValueError("K must be a multiple of 4")
        if self.M * self.N >= 2 ** 32 or self.Mg * self.K >= 2 ** 32 or self.K * self.N >= 2 ** 32:
            raise ValueError("tensors are indexed with u32 elements (shape.wgsl:10-33): each must have < 2^32 elements")

    # -- sizes -------------------------------------------------------------------------------------------------
    @property
    def Mg(self) -> int:  # rows per rank
        return self.M // self.world

    @property
    def np_(self) -> int:  # columns per panel (uniform plans only)
        if self.panel_cols is not None:
            raise ValueError("ragged plan: use cols_of(panel)")
        return self.N // self.npanels

    @property
    def panel_elems(self) -> int:  # one rank's slot of one panel (uniform plans only)
        return self.Mg * self.np_

    def cols_of(self, panel: int) -> int:
        return self.panel_cols[panel] if self.panel_cols is not None else self.N // self.npanels

    def col0_of(self, panel: int) -> int:  # first column of a panel
        return sum(self.panel_cols[:panel]) if self.panel_cols is not None else panel * (self.N // self.npanels)

    def slot_elems_of(self, panel: int) -> int:  # one rank's slot of this panel
        return self.Mg * self.cols_of(panel)

    def gathered_elems(self) -> int:
        return self.M * self.N

    def a_rows(self, rank: int):
        return rank * self.Mg, self.Mg

    # -- views (units: elements), all in the reference's ViewShape model --------------------------------------------
    def a_shape(self) -> ViewShape:  # this rank's A_g, its own dense tensor
        return ViewShape((self.Mg, self.K, 1), self.Mg, self.Mg * self.K, 0)

    def b_panel_shape(self, panel: int) -> ViewShape:  # B[:, panel columns] == GpuMatrix::columns (tensor.rs:596-610)
        return ViewShape((self.K, self.cols_of(panel), 1), self.K, self.K * self.N, self.col0_of(panel) * self.K)

    def _panel_start(self, panel: int) -> int:  # the gathered buffer is [panel][rank][cols * Mg]: M elements per column
        return self.col0_of(panel) * self.M

    def out_shape(self, panel: int, rank: int) -> ViewShape:  # where rank's GEMM of `panel` writes inside the gathered buffer
        se = self.slot_elems_of(panel)
        return ViewShape((self.Mg, self.cols_of(panel), 1), self.Mg, se, self._panel_start(panel) + rank * se)

    def panel_range(self, panel: int):  # contiguous element range of a panel in the gathered buffer (the all-gather's output)
        return self._panel_start(panel), self.world * self.slot_elems_of(panel)

    def cube_shape(self, panel: int) -> ViewShape:  # the gathered panel as a GpuCube [Mg, np, P]
        se = self.slot_elems_of(panel)
        return ViewShape((self.Mg, self.cols_of(panel), self.world), self.Mg, se, self._panel_start(panel))

    def panel_of_col(self, col: int):
        """(panel, column within the panel) of global column `col`."""
        if self.panel_cols is None:
            return divmod(col, self.N // self.npanels)
        p = 0
        while col >= self.panel_cols[p]:
            col -= self.panel_cols[p]
            p += 1
        return p, col

    def element_index(self, row: int, col: int) -> int:
        """Flat index of C[row, col] in the gathered buffer."""
        g, i = divmod(row, self.Mg)
        p, j = self.panel_of_col(col)
        return self._panel_start(p) + g * self.slot_elems_of(p) + j * self.Mg + i


class ShardedGemm:
    """One rank of the M-sharded GEMM.

    local_gemm(out_shape, a_shape, b_shape)  enqueues C_g[:, panel] = A_g * B[:, panel] into the gathered buffer
    all_gather(start, count_per_rank, rank)  starts the (possibly asynchronous) in-place all-gather of the contiguous
                                             range [start, start + world*count_per_rank); returns a handle or None
    wait(handle)                             blocks the compute stream on that collective
    """

    def __init__(self, plan: MShardPlan, rank: int, local_gemm: Callable, all_gather: Callable, wait: Callable = lambda h: None,
                 always_gather: bool = False):
        if not 0 <= rank < plan.world:
            raise ValueError("rank out of range")
        self.plan, self.rank = plan, rank
        self._gemm, self._gather, self._wait = local_gemm, all_gather, wait
        self._always_gather = always_gather  # run the collective even with one rank (exercises the plumbing)

    def step(self) -> None:
        """All panels: GEMM of panel i, then start its all-gather; the collectives drain while later panels compute."""
        pl = self.plan
        handles: List = []
        for p in range(pl.npanels):
            self._gemm(pl.out_shape(p, self.rank), pl.a_shape(), pl.b_panel_shape(p))
            if pl.world > 1 or self._always_gather:
                start, _ = pl.panel_range(p)
                handles.append(self._gather(start, pl.slot_elems_of(p), self.rank))
        for h in handles:
            self._wait(h)


def tapered_panels(N: int, main_cols: int, ratio: float, tile: int = 256, max_tail: int = 8) -> List[int]:
    """Widths of the N-panels with a tapered tail: equal panels of `main_cols`, then panels that shrink by `ratio` (the time of a panel's exchange
    over the time of its Gemm, < 1; rounded UP to whole tiles so that every exchange still hides under the next panel's Gemm) down to one tile
    column -- the only exposed exchange of a step is then that of a one-tile-wide panel instead of a full one. At most `max_tail` panels follow
    the equal ones (what the one-launch kernels take, wg_gemm_sharded_panels); the last panel takes whatever is left of N.
    Pure host arithmetic (tests/test_dist_gloo.py)."""
    if N <= 0 or main_cols <= 0 or main_cols % tile or not 0.0 < ratio < 1.0:
        raise ValueError("tapered_panels: N, main_cols (a multiple of the tile) > 0 and 0 < ratio < 1")
    taper, w = [], main_cols // tile
    while w > 1 and len(taper) < max_tail - 1:
        w = max(1, min(w - 1, -(-int(w * ratio * 1000) // 1000)))  # ceil(w * ratio), but strictly narrower
        taper.append(w * tile)
    if not taper or taper[-1] != tile:  # a last panel of one tile column closes the taper
        taper.append(tile)
    tail_total = sum(taper)
    if N <= tail_total + main_cols:  # not enough columns for equal panels in front of a taper: a plain split
        n = max(1, -(-N // main_cols))
        return [main_cols] * (n - 1) + [N - (n - 1) * main_cols]
    n_main = (N - tail_total) // main_cols
    left = N - n_main * main_cols - tail_total  # < main_cols: one more panel, placed where the widths stay in descending order
    frag, whole = left % tile, left - left % tile
    tail = list(taper)
    if whole:
        at = 0
        while at < len(tail) and tail[at] > whole:
            at += 1
        tail.insert(at, whole)
    while len(tail) > max_tail:  # too many for the one-launch kernels: join the two narrowest panels in front of the last one
        tail[-3:-1] = [tail[-3] + tail[-2]]
    tail[-1] += frag  # a ragged N: the fraction of a tile goes to the very last panel
    return [main_cols] * n_main + tail


# ---------------------------------------------------------------------------------------------------------------------
# product path: the C ABI's communicator and M-sharded Gemm
# ---------------------------------------------------------------------------------------------------------------------
class GatherMode:
    RCCL = 0       # staging cube + in-place ncclAllGather per panel + relayout into C (wg_gather_mode WG_GATHER_RCCL)
    NONE = 2
    PEER_STAGED = 3  # Gemm into a staging cube, one contiguous copy per peer link + flag, wait kernel + relayout on the receiver (WG_GATHER_PEER_STAGED)


def new_unique_id() -> bytes:
    """ncclGetUniqueId through the C ABI (rank 0 calls this and ships the bytes to the other ranks out of band)."""
    from . import _lib
    buf = ctypes.create_string_buffer(_lib.WG_COMM_ID_BYTES)
    _lib.check(_lib.lib.wg_comm_unique_id(buf))
    return buf.raw


class Comm:
    """wg_comm: one rank of a group of GpuInstances (one per GPU). `unique_id=None` creates a communicator without a collective
    library (peer copies only; the caller brings its own barrier -- e.g. two ranks sharing one GPU in the tests)."""

    def __init__(self, gpu, nranks: int, rank: int, unique_id: Optional[bytes]):
        from . import _lib
        self._lib, self.gpu, self.nranks, self.rank = _lib, gpu, nranks, rank
        h = ctypes.c_void_p()
        idbuf = ctypes.create_string_buffer(unique_id, _lib.WG_COMM_ID_BYTES) if unique_id is not None else None
        _lib.check(_lib.lib.wg_comm_create(gpu._ctx.handle, nranks, rank, idbuf, ctypes.byref(h)))
        self._h = h
        self._peers = {}  # id(tensor) -> (ctypes array of wg_buf*, keepalive)

    def close(self):
        if getattr(self, "_h", None):
            for k, v in self._peers.items():
                if k == "__stages_key__":
                    continue
                for b in v[1]:
                    self._lib.lib.wg_buf_destroy(b)
            self._peers = {}
            if self.gpu._ctx.handle:
                self._lib.lib.wg_comm_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def reported_size(self) -> int:
        """ncclCommCount: the rank count the collective library itself reports (0 for a communicator without one)."""
        n = ctypes.c_int(0)
        self._lib.check(self._lib.lib.wg_comm_reported_size(self._h, ctypes.byref(n)))
        return int(n.value)

    @property
    def has_collectives(self) -> bool:
        return bool(self._lib.lib.wg_comm_has_collectives(self._h))

    @property
    def bytes_sent(self) -> int:
        return int(self._lib.lib.wg_comm_bytes_sent(self._h))

    def all_gather(self, tensor, first_elem: int, elems_per_rank: int) -> None:
        from .wgcore import wg_dtype
        self._lib.check(self._lib.lib.wg_all_gather(self._h, wg_dtype(tensor.dtype), tensor._h, first_elem, elems_per_rank))

    def set_pipelined(self, on: bool) -> None:
        """Staged engine: defer each call's last panel (wait + relayout) to the next call / join(), hiding its exchange under the next step."""
        self._lib.check(self._lib.lib.wg_comm_set_pipelined(self._h, 1 if on else 0))

    def set_one_launch(self, on) -> None:
        """f16 products of >= one round of tiles as ONE kernel over all N-panels with per-panel arrival counters: True / False force it / the
        panel-by-panel launches, None = by engine (the default: RCCL on, staged off)."""
        self._lib.check(self._lib.lib.wg_comm_set_one_launch(self._h, -1 if on is None else (1 if on else 0)))

    def set_wait_timing(self, on: bool) -> None:
        """Diagnostics: stamp the compute stream around every wait for a panel's exchange (wg_comm_set_wait_timing); read with wait_times()."""
        self._lib.check(self._lib.lib.wg_comm_set_wait_timing(self._h, 1 if on else 0))

    def wait_times(self, capacity: int = 4096):
        """[(panel, ms the compute stream waited for that panel's exchange), ...] since the last call, oldest first (synchronises the context)."""
        pan, ms, n = (ctypes.c_uint32 * capacity)(), (ctypes.c_float * capacity)(), ctypes.c_uint32(0)
        self._lib.check(self._lib.lib.wg_comm_wait_times(self._h, pan, ms, capacity, ctypes.byref(n)))
        return [(int(pan[i]), float(ms[i])) for i in range(n.value)]

    def join(self) -> None:
        self._lib.check(self._lib.lib.wg_comm_join(self._h))

    def flush(self) -> None:
        self._lib.check(self._lib.lib.wg_comm_flush(self._h))

    def barrier(self) -> None:
        self._lib.check(self._lib.lib.wg_comm_barrier(self._h))

    # -- staged peer copies (GatherMode.PEER_STAGED): the communicator's staging cubes + flag array ---------------------------
    def stage_reserve(self, nbytes: int):
        """Make the staging cubes (>= 2 * itemsize * M * N bytes) and the flag array exist; returns their two wg_buf handles."""
        st, fl = ctypes.c_void_p(), ctypes.c_void_p()
        self._lib.check(self._lib.lib.wg_comm_stage_reserve(self._h, int(nbytes), ctypes.byref(st), ctypes.byref(fl)))
        self._stage = (st, fl)
        return st, fl

    def stage_export(self, nbytes: int):
        """(stage handle bytes, flags handle bytes) to ship to every peer."""
        st, fl = self.stage_reserve(nbytes)
        out = []
        for b in (st, fl):
            buf = ctypes.create_string_buffer(self._lib.WG_IPC_HANDLE_BYTES)
            self._lib.check(self._lib.lib.wg_buf_ipc_export(b, buf))
            out.append(buf.raw)
        return tuple(out)

    def set_peer_stages(self, handles) -> None:
        """handles[r] = rank r's stage_export() pair (handles[self.rank] is ignored). Idempotent: the same pairs again (a second workload on
        the same communicator) keep the existing mappings -- an allocation is opened once per process."""
        key = tuple(tuple(p) if p is not None else None for p in handles)
        if self._peers.get("__stages_key__") == key:
            return
        old = self._peers.pop("__stages__", None)
        if old is not None:  # the peers re-allocated their cubes: drop the stale mappings
            for b in old[1]:
                self._lib.lib.wg_buf_destroy(b)
        sa, fa = (ctypes.c_void_p * self.nranks)(), (ctypes.c_void_p * self.nranks)()
        opened = []
        for r, pair in enumerate(handles):
            if r == self.rank:
                sa[r], fa[r] = self._stage[0].value, self._stage[1].value
                continue
            for arr, hb in ((sa, pair[0]), (fa, pair[1])):
                b = ctypes.c_void_p()
                self._lib.check(self._lib.lib.wg_buf_ipc_open(self.gpu._ctx.handle, ctypes.create_string_buffer(hb, self._lib.WG_IPC_HANDLE_BYTES), ctypes.byref(b)))
                arr[r] = b.value
                opened.append(b)
        self._lib.check(self._lib.lib.wg_comm_set_peer_stages(self._h, sa, fa))
        self._peers["__stages__"] = ((sa, fa), opened)
        self._peers["__stages_key__"] = key

    def set_local_peer_stages(self, comms) -> None:
        """The ranks share this process (tests): comms[r] is rank r's Comm (each already stage_reserve()d)."""
        sa, fa = (ctypes.c_void_p * self.nranks)(), (ctypes.c_void_p * self.nranks)()
        for r, cm in enumerate(comms):
            sa[r], fa[r] = cm._stage[0].value, cm._stage[1].value
        self._lib.check(self._lib.lib.wg_comm_set_peer_stages(self._h, sa, fa))
        self._peers["__stages__"] = ((sa, fa), [])

    def sharded_gemm(self, out, a_rows, b, variant=0, mode: int = GatherMode.RCCL, panel_cols=0) -> None:
        """out (M x N GpuMatrix, on every rank) = op(A) * B with A sharded on M: `a_rows` is this rank's row block (M/P x K, or
        K x M/P for the GemmTr variants), `b` (K x N) replicated. Views or tensors; enqueues and returns (see wg_gemm_sharded for
        when `out` is complete). `panel_cols`: one width for every N-panel (0 = default), or the list of the panels' widths
        (wg_gemm_sharded_panels: a tapered tail, see `tapered_panels`)."""
        from .wgcore import as_view, wg_dtype
        ov, av, bv = as_view(out), as_view(a_rows), as_view(b)
        if not isinstance(panel_cols, int):
            widths = (ctypes.c_uint32 * len(panel_cols))(*[int(w) for w in panel_cols])
            self._lib.check(self._lib.lib.wg_gemm_sharded_panels(self._h, int(variant), wg_dtype(ov.dtype), int(mode), widths, len(panel_cols), ov.buffer()._h,
                                                                 ov.shape().to_c(), av.buffer()._h, av.shape().to_c(), bv.buffer()._h, bv.shape().to_c()))
            return
        self._lib.check(self._lib.lib.wg_gemm_sharded(self._h, int(variant), wg_dtype(ov.dtype), int(mode), int(panel_cols), ov.buffer()._h, ov.shape().to_c(),
                                                      av.buffer()._h, av.shape().to_c(), bv.buffer()._h, bv.shape().to_c()))
