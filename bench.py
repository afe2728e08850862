#!/usr/bin/env python3
"""bench.py -- throughput of the wgebra dense hot path on MI355X, against its roofline, with the CPU port beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
one Gemm / Gemv / batched Reduce dispatch (for N > 1: the M-sharded Gemm of every rank + the RCCL all-gather of C).
Rank 0 prints ONE JSON line (schema: see DESIGN.md "Measurement").

Workloads (BASELINE.json `configs`):
  gemm_f32_4096         f32 GEMM 4096^3           (configs[1])   MFMA-bound   TFLOP/s
  gemm_f16_8192         f16 GEMM 8192^3           (configs[2])   MFMA-bound   TFLOP/s
  gemv_f32_4096x65536   f32 GEMV 4096 x 65536     (configs[3])   HBM-bound    GB/s
  gemvtr_f32_65536x4096 the transposed twin       (configs[3])   HBM-bound    GB/s
  reduce_f32_4096x65536 4096 Sum-reductions of 65536 (configs[3]) HBM-bound   GB/s
  gemm_f16_32768        f16 GEMM 32768^3, M-sharded over the ranks + all-gather (configs[4])
With --gpus N > 1 every GEMM workload is M-sharded over the N ranks ("strong": the total problem is fixed).

Only the cpu_baseline leg touches oracle/ (as the thing timed on the host cores, never as the product path).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# chip peaks: /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters" (dense, no sparsity)
PEAK_HBM_GBS = 8000.0
PEAK_MFMA_TFLOPS = {"f32": 157.3, "f16": 2500.0}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def box_facts() -> dict:
    """What tells one box of the pool from another WITHOUT touching the GPU: the first amdgpu card in sysfs that reports a chip id (its unique id, VBIOS, and the
    MEC / SDMA / SMC / RLC firmware versions) and the kernel's release. The replayed config-1 figure falls into two populations across boxes (2.6 vs 3.5 us,
    STATUS.md): these are the fields to hold it against."""
    out = {}
    try:
        base = "/sys/class/drm"
        cards = sorted((c for c in os.listdir(base) if c.startswith("card") and c[4:].isdigit()), key=lambda c: int(c[4:]))
        for c in cards:
            dev = os.path.join(base, c, "device")
            if not os.path.exists(os.path.join(dev, "unique_id")):
                continue
            for name, key in (("unique_id", "chip_id"), ("vbios_version", "vbios"), ("fw_version/mec_fw_version", "fw_mec"), ("fw_version/sdma_fw_version", "fw_sdma"),
                              ("fw_version/smc_fw_version", "fw_smc"), ("fw_version/rlc_fw_version", "fw_rlc")):
                try:
                    with open(os.path.join(dev, name)) as f:
                        out[key] = f.read().strip()
                except OSError:
                    pass
            break
    except OSError:
        pass
    try:
        with open("/proc/sys/kernel/osrelease") as f:
            out["kernel"] = f.read().strip()
    except OSError:
        pass
    return out


# The driver keeps and parses the LAST stdout line; round 5's 25 KB line (27 workloads' full objects under "others") came back unparsed.
# The line is therefore compact -- the contract's keys, `config` with the flat C1-C5 scalars, `roofline`, `cpu_baseline`, `checks`, `targets` --
# and everything bulky ("others", the per-rank arrays of "ranks_detail" when they are long) goes to a sidecar file (--detail) and to stderr.
MAX_LINE_BYTES = 8192
DEFAULT_DETAIL = os.path.join(ROOT, "gpurun_out", "bench_detail.json")


def finite(o):
    """`o` with every non-finite float replaced by None: the line and the sidecar are STRICT JSON (json.dumps(..., allow_nan=False))."""
    if isinstance(o, dict):
        return {str(k): finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [finite(v) for v in o]
    if isinstance(o, (float, np.floating)):
        return float(o) if np.isfinite(o) else None
    if isinstance(o, np.integer):
        return int(o)
    if isinstance(o, np.bool_):
        return bool(o)
    return o


def compact_line(full: dict) -> dict:
    """The stdout line out of the full record: without "others"; "ranks_detail" only while it is short; and, should the rest still exceed
    MAX_LINE_BYTES (it is ~4.5 KB at N = 1), without `config`'s copies of the scalars that `targets` carries anyway."""
    line = {k: v for k, v in full.items() if k != "others"}
    if line.get("ranks_detail") is not None and len(json.dumps(finite(line["ranks_detail"]), allow_nan=False)) > 1536:
        line["ranks_detail"] = None
    size = lambda: len(json.dumps(finite(line), allow_nan=False))  # noqa: E731
    if size() > MAX_LINE_BYTES:
        line["ranks_detail"] = None
    if size() > MAX_LINE_BYTES and isinstance(line.get("targets"), dict):
        line["config"] = {k: v for k, v in line["config"].items() if k not in line["targets"]}
    if size() > MAX_LINE_BYTES:
        for k in ("data", "checks"):
            line[k] = None if k == "checks" else str(line[k])[:60]
    return line


def emit(full: dict, detail_path, to_stdout) -> str:
    """Writes the full record (strict JSON) to `detail_path` (None/"-" = no file) and to stderr, then hands the compact line to `to_stdout`.
    Returns the compact line's text."""
    full = finite(full)
    line = compact_line(full)
    if detail_path and detail_path != "-":
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                f.write(json.dumps(full, allow_nan=False) + "\n")
            line["detail"] = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT + os.sep) else detail_path
        except OSError as e:  # a read-only tree must not take the line down
            log(f"[bench] could not write the detail sidecar {detail_path}: {e}")
            line["detail"] = None
    else:
        line["detail"] = None
    if "targets" in line:
        line["targets"] = line.pop("targets")  # stays the LAST key (the driver's `tail` keeps the last 2000 characters of stdout)
    log("[bench detail] " + json.dumps(full, allow_nan=False))
    text = json.dumps(line, allow_nan=False)
    assert len(text) <= MAX_LINE_BYTES, f"bench line is {len(text)} bytes (> {MAX_LINE_BYTES}): move more of it to the sidecar"
    to_stdout(text)
    return text


def usable_cpus() -> int:
    """Host threads this process can actually run at once: the scheduler affinity mask, capped by the cgroup's CPU quota (a container on a
    256-thread host with a quota of 8 CPUs runs 8 threads' worth however many it starts -- 256 OpenMP threads there spend their time in
    fork/join, which is what round 3's 120 ms "runs" of a 512 KiB Gemv slice measured)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:  # cgroup v1
                q, per = int(f.read()), int(g.read())
                if q > 0 and per > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.999)))
    return max(1, n)


def time_port(C, run, work_items: int, budget_s: float):
    """Times `run()` (one pass of the CPU port over its sample) on 1 thread and on min(work items, usable CPUs) threads, half the budget each, and
    returns the faster: (seconds per run, threads used, runs timed). A port that is slower on many threads than on one (a sample too small to
    share) is reported as what it is: a one-thread number."""
    cands = sorted({1, max(1, min(int(work_items), usable_cpus()))})
    best = None
    for n in cands:
        C.set_num_threads(n)
        run()  # warm (page faults, thread pool)
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s / len(cands) or reps < 3:
            run()
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        if best is None or dt < best[0]:
            best = (dt, n, reps)
    return best


# ------------------------------------------------------------------------------------------------------------
# synthetic data: seeded, U[-1,1) (full-range random operands: zero/constant fills flatter the clocks)
# ------------------------------------------------------------------------------------------------------------
# Operand distribution (SURVEY 8(d) "inputs"): "pm1" = U[-1, 1) (the headline: sign bits toggle, the harder case for a power-capped part);
# "u01" = U[0, 1), the reference's own (`GpuTensor::new_random`, gemm.rs:152 / nalgebra `new_random`) -- reported next to it as *_u01 workloads and
# `c3_*_u01` / `c5_*_u01` targets; "zero" = experiments only (zero operands take the chip off its power cap: kernel time then measures cycles).
VALUES = os.environ.get("WG_BENCH_VALUES", "pm1")


class values_as:
    """`with values_as("u01"):` -- the distribution rand_block draws from inside the block."""

    def __init__(self, v):
        self.v = v

    def __enter__(self):
        global VALUES
        self.saved, VALUES = VALUES, (self.v or VALUES)

    def __exit__(self, *exc):
        global VALUES
        VALUES = self.saved


def rand_block(seed: int, n: int, dtype) -> np.ndarray:
    rng = np.random.default_rng(seed)
    if VALUES == "u01":
        return rng.random(n, dtype=np.float32).astype(dtype)
    if VALUES == "zero":
        return np.zeros(n, dtype)
    return (rng.random(n, dtype=np.float32) * np.float32(2) - np.float32(1)).astype(dtype)


def device_random(wg, gpu, shape, dtype, seed):
    """A tensor of `shape` filled by tiling a 16 Mi-element seeded random block with device-side copies
    (keeps host RNG time and PCIe traffic bounded for the multi-GiB configs; every element is still random data)."""
    n = int(np.prod(shape))
    block_n = min(n, 1 << 24)
    S = wg.BufferUsages
    t = wg.TensorBuilder.tensor(shape, S.STORAGE | S.COPY_SRC | S.COPY_DST).build(gpu.device(), dtype)
    blk = wg.TensorBuilder.vector(block_n, S.STORAGE | S.COPY_SRC).build_init(gpu.device(), rand_block(seed, block_n, dtype))
    item = np.dtype(dtype).itemsize
    from wgmath_amd._lib import check, lib
    off = 0
    while off < n:
        m = min(block_n, n - off)
        check(lib.wg_buf_copy(gpu._ctx.handle, blk._h, 0, t._h, off * item, m * item))
        off += m
    gpu.sync()
    return t


# ------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------
DIST = None  # set by main() for multi-rank runs: {"comm": Comm, "mode": "rccl"|"rccl_cus4"|"staged", "barrier": fn, "all_gather_object": fn}


def plan_panel_cols(Mg: int, N: int, cus: int, tile: int = 256, target_panels: int = 16) -> int:
    """Width of the N-panels of the sharded Gemm: a whole number of rounds of `cus` workgroups per panel (a Gemm workgroup owns a CU,
    so a panel of 1.3 rounds costs 2), as many panels as that allows up to `target_panels` (panel i's exchange hides under panel
    i+1's Gemm; only the last one is exposed), at least 2 when N allows."""
    tiles_m = max(1, -(-Mg // tile))
    tcols = max(1, N // tile)
    best = None
    for c in range(1, tcols + 1):
        npan = -(-tcols // c)
        if npan < 2 and tcols >= 2:
            break
        t = tiles_m * c
        waste = (-(-t // cus)) * cus / t  # launched rounds / useful rounds
        score = (round(waste, 3), abs(npan - target_panels))
        if best is None or score < best[0]:
            best = (score, c)
    c = best[1] if best else tcols
    return min(N, c * tile)


class Workload:
    name = ""
    dtype = "f32"
    metric = ""
    unit = ""
    bound = "hbm"
    kernel = ""  # dominant kernel, for the rocprof cross-check

    def setup(self, wg, gpu, rank, world): ...
    def step(self): ...
    def units_per_step(self) -> float: ...        # whole job (all ranks), in `unit`-numerator units (flops or bytes)
    def algorithmic_per_launch(self) -> float: ...  # this rank's dominant kernel, per launch
    def cpu_baseline(self, budget_s: float) -> dict: ...
    def check(self) -> None: ...                  # cheap sanity check of the result (not timed)

    def algorithmic_bytes(self) -> float:
        """Compulsory HBM bytes of one launch of the dominant kernel (operands read once, result written once): what the measured
        `roofline.traffic` is compared with. HBM-bound workloads: their algorithmic bytes per launch."""
        return self.algorithmic_per_launch()


class GemmWorkload(Workload):
    bound = "mfma"
    metric = "gemm_tflops"
    unit = "TFLOP/s"

    def __init__(self, name, M, N, K, dtype, trans=False, values=None):
        self.name, self.M, self.N, self.K = name, M, N, K
        self.values = values  # operand distribution of THIS workload (None: the run's, --dist)
        self.dtype = dtype
        self.trans = trans  # GemmTr: m1 is stored K x M (op(A) = m1^T)
        self.np_dtype = np.float32 if dtype == "f32" else np.float16
        self.kernel = "gemm_f32_kernel" if dtype == "f32" else "gemm_f16_m16_kernel"
        # mid-size outputs: f16 runs the 128 x 128 kernel; f32 the mid-size tile family (no K cut since round 4)
        if dtype == "f16" and (M // 256) * (N // 256) < 256 and M * N <= 2560 * 2560:
            self.kernel = "gemm_f16_t128_kernel"
        if dtype == "f32" and (M // 256) * (N // 128) < 256 and M == N == K == 2048:
            self.kernel = "gemm_f32_mid_kw_kernel" if not trans else "gemm_f32_mid_kernel"  # the mid-size tile family: 64 x 64 k-split tiles (GemmTr: 128 x 64)

    def setup(self, wg, gpu, rank, world):
        self.wg, self.gpu, self.rank, self.world = wg, gpu, rank, world
        # M-shard: rank g owns A_g = A[g*Mg:(g+1)*Mg, :] as its own contiguous col-major tensor, B is replicated, every rank ends with
        # the full M x N column-major C (wgmath_amd/csrc/comm.hip). DIST is set by main() when the run has a communicator.
        self.dist = DIST if (DIST is not None and DIST.get("comm") is not None) else None
        if self.M % (4 * world):
            raise ValueError(f"M={self.M} does not split into {world} vec4-aligned row blocks")
        self.Mg = self.M // world
        with values_as(self.values):
            self.A = device_random(wg, gpu, (self.K, self.Mg) if self.trans else (self.Mg, self.K), self.np_dtype, 0xA000 + rank)
            self.B = device_random(wg, gpu, (self.K, self.N), self.np_dtype, 0xB000)  # replicated
        S = wg.BufferUsages
        self.C = wg.TensorBuilder.matrix(self.M, self.N, S.STORAGE | S.COPY_SRC | S.COPY_DST).build(gpu.device(), self.np_dtype)
        self.gemm = wg.Gemm.from_device(gpu.device())
        self.shapes = wg.ViewShapeBuffers()
        self.enc = gpu.device().create_command_encoder()
        self.pass_ = self.enc.compute_pass("bench", None)
        self.variant = wg.GemmVariant.GemmTr if self.trans else wg.GemmVariant.Gemm
        self.mode, self.panel_cols, self.npanels = None, 0, 1
        if self.dist is not None:
            self.set_mode(self.dist["mode"])

    def set_mode(self, mode):
        """Exchange engine of the sharded run: "rccl" (staging cube + ncclAllGather + relayout; "rccl_cus4" is the same engine with 4 instead of 8
        CUs left to RCCL's kernels) or "staged" (contiguous copies on one SDMA engine per link + relayout)."""
        from wgmath_amd.sharded import GatherMode
        comm, world = self.dist["comm"], self.world
        self.mode = mode
        rccl = mode.startswith("rccl")
        cus = int(self.gpu.adapter().get("stream_compute_units") or self.gpu.adapter()["compute_units"])
        if rccl and self.dtype == "f16":
            cus = int(self.gpu.adapter()["compute_units"])  # ONE scheduler-driven launch per step: panels need not be whole rounds of the masked stream
        self.panel_cols = plan_panel_cols(self.Mg, self.N, cus, tile=256 if self.dtype == "f16" else 128)
        self.npanels = -(-self.N // self.panel_cols)
        self.panel_widths = None
        # Tapered tail (wg_gemm_sharded_panels) for the engine that runs a step as ONE kernel: equal panels, then narrower and narrower ones, each at
        # least `taper` times the one before it (taper = exchange time / Gemm time of a panel, ~0.7 at 60 GB/s per link and 1.4 PFLOP/s whatever the
        # rank count: both scale with the rows per rank) -- every exchange still hides under the next panel's Gemm and the one exchange nothing
        # hides is that of a single tile column (1/8 of a panel). WG_BENCH_TAPER=0 switches it off, another value sets the ratio.
        taper = float(os.environ.get("WG_BENCH_TAPER", "0.72"))
        if rccl and self.dtype == "f16" and taper > 0 and self.panel_cols % 256 == 0 and self.npanels > 2:
            from wgmath_amd.sharded import tapered_panels
            self.panel_widths = tapered_panels(self.N, self.panel_cols, taper)
            self.npanels = len(self.panel_widths)
        if world > 1:
            # the panel plan is part of the slot layout of the staging cubes: every rank must cut N the same way (WG_BENCH_TAPER / panel planning read per rank).
            # Nothing in the library compares plans across ranks (wgebra_hip.h, wg_gemm_sharded_panels): compared here, once, before the first step.
            plans = self.dist["all_gather_object"](tuple(self.panel_widths or [self.panel_cols]))
            if any(p != plans[0] for p in plans):
                raise ValueError(f"the ranks planned different N-panels: {plans}")
        self.gather_mode = GatherMode.PEER_STAGED if mode == "staged" else GatherMode.RCCL
        if rccl and self.dtype == "f16":  # the one-launch form defers its last panel too (two cubes by step parity inside the library)
            comm.set_pipelined(os.environ.get("WG_BENCH_PIPELINED", "1") != "0")
        if mode == "staged":  # two staging cubes (step parity) + flag array, exported to every peer
            pair = comm.stage_export(2 * self.M * self.N * np.dtype(self.np_dtype).itemsize)
            if world > 1:
                comm.set_peer_stages(self.dist["all_gather_object"](pair))
            # pipelined steps: a step's last panel (the one exchange nothing of its own step hides) completes under the next step's first
            # Gemm; finish() joins, inside the timed region, so every step's C is complete when the clock stops
            comm.set_pipelined(os.environ.get("WG_BENCH_PIPELINED", "1") != "0")

    def step(self):
        if self.dist is None:
            self.gemm.dispatch_generic(self.gpu.device(), self.shapes, self.pass_, self.C, self.A, self.B, self.variant)
            return
        self.dist["comm"].sharded_gemm(self.C, self.A, self.B, int(self.variant), self.gather_mode, self.panel_widths or self.panel_cols)

    def finish(self):
        """End of a run of steps (inside the timed region): complete what pipelined steps deferred."""
        if self.dist is not None:
            self.dist["comm"].join()

    def units_per_step(self):
        return 2.0 * self.M * self.N * self.K

    def algorithmic_bytes(self):
        es = 4 if self.dtype == "f32" else 2
        return float(es) * (self.Mg * self.K + self.K * self.N + self.Mg * self.N) / self.npanels

    def algorithmic_per_launch(self):
        return 2.0 * self.Mg * self.N * self.K / self.npanels  # mean over the launches of a step (one launch = one N-panel of this rank's row block)

    def launches_per_step(self):
        return self.npanels

    def gather_bytes_per_step(self):
        """Payload this rank receives per step from the other ranks (the all-gather of C)."""
        return (self.world - 1) * self.Mg * self.N * np.dtype(self.np_dtype).itemsize

    def check(self):
        # sampled entries of C against f64 on the host: this rank's own rows always; with a communicator also rows the OTHER ranks
        # computed and pushed / gathered here (their A blocks are regenerated from the seed: device_random is deterministic)
        gpu = self.gpu
        rng = np.random.default_rng(1)
        cols = np.unique(rng.integers(0, self.N, 24))
        item = np.dtype(self.np_dtype).itemsize
        from wgmath_amd._lib import check, lib

        def read_range(t, start, n):
            out = np.empty(n, self.np_dtype)
            check(lib.wg_buf_read(gpu._ctx.handle, t._h, start * item, out.ctypes.data, n * item))
            return out

        Bc = np.stack([read_range(self.B, c * self.K, self.K) for c in cols], axis=1).astype(np.float64)  # K x ncols
        owners = [self.rank] if (self.dist is None or self.world == 1) else sorted({self.rank, (self.rank + 1) % self.world, (self.rank - 1) % self.world})
        for g in owners:
            rows = np.unique(rng.integers(0, self.Mg, 6))
            if g == self.rank:
                A = self.A.read(gpu.device())
            else:  # the peer's row block, regenerated the way device_random filled it (one seeded block, tiled)
                n = self.Mg * self.K
                with values_as(self.values):
                    blk = rand_block(0xA000 + g, min(n, 1 << 24), self.np_dtype)
                A = np.resize(blk, n)
            A = (A.reshape(self.K, self.Mg, order="F").T if self.trans else A.reshape(self.Mg, self.K, order="F"))[rows].astype(np.float64)
            got = np.empty((rows.size, cols.size))
            for jc, c in enumerate(cols):
                colbuf = read_range(self.C, int(c) * self.M + g * self.Mg, self.Mg)
                got[:, jc] = colbuf[rows]
            truth, sabs = A @ Bc, np.abs(A) @ np.abs(Bc)
            tol = 2 * np.sqrt(self.K) * 2.0 ** -24 * sabs + (2.0 ** -11 * np.abs(truth) + 2.0 ** -25 if self.dtype == "f16" else 0)
            err = np.abs(got - truth)
            assert (err <= tol).all(), f"bench sanity check failed (rows of rank {g} as seen on rank {self.rank}): worst err/tol {(err / tol).max():.3g}"
            # the same sample in ulps of the result's own format at the true value (the figure the north star asks to be stated; tests/test_gpu_ulp.py
            # asserts the bounds, against the restated WGSL order as well)
            # f32: ulps of sum|a||b| (the operands are U[-1,1): results cancel towards 0, "ulps of the result" is unbounded for ANY summation order);
            # f16: ulps of the result -- its one rounding dominates (the f32 accumulation's share is allowed for in the assertion above, not here)
            scale = sabs if self.dtype == "f32" else np.abs(truth)
            ulp = np.spacing(scale.astype(self.np_dtype)).astype(np.float64)
            self.max_ulp_vs_f64 = max(getattr(self, "max_ulp_vs_f64", 0.0), float((err / ulp).max()))

    def cpu_baseline(self, budget_s):
        # The reference's only GEMM is the f32 WGSL kernel; its CPU port is timed on a slice with the SAME K
        # (for the f16 workloads this is still the f32 port: there is no reference f16 path to port).
        from oracle import wgsl_oracle as wo
        C = wo.CLib()
        K = self.K
        Ms = min(self.M, 8192)
        Ns = min(self.N, 256 if K > 8192 else 1024)
        a, b = rand_block(1, Ms * K, np.float32), rand_block(2, K * Ns, np.float32)
        out = np.zeros(Ms * Ns, np.float32)
        s1, s2, so = wo.Shape(Ms, K), wo.Shape(K, Ns), wo.Shape(Ms, Ns)
        active_wgs = -(-(Ms // 4) // 64)  # gemm.wgsl:86: only invocations x < M/4 do work
        flops_per_wg = 2.0 * min(256, Ms) * Ns * K  # 64 invocations x 4 rows
        threads = max(1, min(active_wgs, usable_cpus()))  # one OpenMP task per workgroup: more threads than workgroups or CPUs only adds fork/join
        C.set_num_threads(1)
        t0 = time.perf_counter()
        C.gemm(wo.GEMM, out, so, a, s1, b, s2, 0, 1)
        t1 = time.perf_counter() - t0  # one workgroup on one thread
        C.set_num_threads(threads)
        n = int(max(threads, min(active_wgs, threads * budget_s / max(t1, 1e-3))))
        n = min(active_wgs, (n // threads) * threads)  # whole rounds of the threads
        t0 = time.perf_counter()
        C.gemm(wo.GEMM, out, so, a, s1, b, s2, 0, n)
        dt = time.perf_counter() - t0
        v_many, v_one = flops_per_wg * n / dt / 1e12, flops_per_wg / t1 / 1e12
        value, cores, took = (v_many, threads, dt) if v_many >= v_one else (v_one, 1, t1)
        return {"value": value, "unit": "TFLOP/s", "cores": cores, "kind": "port",
                "sample": f"oracle/wgsl_oracle.c `gemm` (f32, naive WGSL order; the reference has no f16 kernel), {min(256, Ms) * (n if cores > 1 else 1)} rows x {Ns} "
                          f"columns x K={K} of the {self.M}x{self.N}x{K} problem, {n if cores > 1 else 1} of {active_wgs} active workgroups on {cores} thread(s) "
                          f"({usable_cpus()} usable CPUs), {took * 1e3:.0f} ms"}


class RowMajorGemmWorkload(GemmWorkload):
    """Gemm / GemmTr on ROW-major views (the reference's `Shape` with row_major_shader_defs(), shape.rs:13-15, shape.wgsl:49-57): wg_gemm_rm. Single GPU only.
    GemmTr: m1 is K x M, m2 K x N, out M x N, all row-major -- in column-major terms a product with BOTH operands contiguous along their output dimension, which
    gemm_f16_nt.hip takes as it lies (f16; f32 transposes m1 into scratch first)."""

    def __init__(self, name, M, N, K, dtype, trans=False):
        super().__init__(name, M, N, K, dtype, trans=trans)
        if dtype == "f16" and trans:
            self.kernel = "gemm_f16_nt_kernel"
        if dtype == "f32" and trans:
            self.kernel = "gemm_f32_kernel"  # <false, true>: the n-contiguous-B tile bodies

    def setup(self, wg, gpu, rank, world):
        if world != 1 or (DIST is not None and DIST.get("comm") is not None):
            raise ValueError("the row-major workloads are single-GPU")
        super().setup(wg, gpu, rank, world)
        self.gemm = wg.Gemm.from_device(gpu.device(), wg.row_major_shader_defs())

        def rm(t, rows, cols):  # a row-major view over the tensor's buffer: index = i * cols + j
            return wg.GpuTensorView(wg.ViewShape([rows, cols, 1], cols, rows * cols, 0), t, 3)
        self.vA = rm(self.A, self.K, self.M) if self.trans else rm(self.A, self.M, self.K)
        self.vB, self.vC = rm(self.B, self.K, self.N), rm(self.C, self.M, self.N)

    def step(self):
        self.gemm.dispatch_generic(self.gpu.device(), self.shapes, self.pass_, self.vC, self.vA, self.vB, self.variant)

    def check(self):
        gpu, rng, item = self.gpu, np.random.default_rng(1), np.dtype(self.np_dtype).itemsize
        from wgmath_amd._lib import check, lib

        def read_range(t, start, n):
            out = np.empty(n, self.np_dtype)
            check(lib.wg_buf_read(gpu._ctx.handle, t._h, start * item, out.ctypes.data, n * item))
            return out
        rows, cols = np.unique(rng.integers(0, self.M, 6)), np.unique(rng.integers(0, self.N, 24))
        B = self.B.read(gpu.device()).reshape(self.K, self.N)[:, cols].astype(np.float64)
        A = self.A.read(gpu.device())
        A = (A.reshape(self.K, self.M)[:, rows].T if self.trans else A.reshape(self.M, self.K)[rows]).astype(np.float64)
        got = np.stack([read_range(self.C, int(r) * self.N, self.N)[cols] for r in rows]).astype(np.float64)
        truth, sabs = A @ B, np.abs(A) @ np.abs(B)
        tol = 2 * np.sqrt(self.K) * 2.0 ** -24 * sabs + (2.0 ** -11 * np.abs(truth) + 2.0 ** -25 if self.dtype == "f16" else 0)
        err = np.abs(got - truth)
        assert (err <= tol).all(), f"bench sanity check failed (row-major): worst err/tol {(err / tol).max():.3g}"
        scale = sabs if self.dtype == "f32" else np.abs(truth)
        self.max_ulp_vs_f64 = float((err / np.spacing(scale.astype(self.np_dtype)).astype(np.float64)).max())


class FewColumnsGemmWorkload(GemmWorkload):
    """f32 Gemm with N <= 64 (a matrix applied to a handful of vectors): HBM-bound on streaming A once, like a GEMV with N right-hand
    sides, so it is reported in GB/s of algorithmic bytes 4 (M K + K N + M N) against the HBM peak (gemm_f32_skinny.hip)."""
    bound = "hbm"
    metric = "gemm_few_columns_gbs"
    unit = "GB/s"

    def __init__(self, name, M, N, K):
        super().__init__(name, M, N, K, "f32")
        self.kernel = "gemm_f32_skinny_kernel"

    def _bytes(self):
        return 4.0 * (self.M * self.K + self.K * self.N + self.M * self.N)

    def units_per_step(self):
        return self._bytes()

    def algorithmic_per_launch(self):
        return self._bytes()

    def cpu_baseline(self, budget_s):
        r = super().cpu_baseline(budget_s)  # the same naive WGSL gemm port, converted to the bytes this workload counts
        r["value"] = r["value"] * 1e12 / (2.0 * self.M * self.N * self.K) * self._bytes() / 1e9
        r["unit"] = "GB/s"
        return r


class GemvWorkload(Workload):
    bound = "hbm"
    metric = "gemv_gbs"
    unit = "GB/s"

    def __init__(self, name, R, C, trans, graph_batch=0, nrhs=1, dtype="f32"):
        self.name, self.R, self.C, self.trans = name, R, C, trans
        self.dtype = dtype  # "f16": this build's extension (f16 elements, f32 accumulation, one rounding); the reference kernel is f32
        self.np_dtype = np.float32 if dtype == "f32" else np.float16
        self.nrhs = nrhs  # right-hand-side columns (`out_ncols` of the reference: grid.y of gemv.wgsl); the matrix is read ONCE for up to 8
        # launch-bound sizes take the one-kernel path (gemv.hip: rows * cols <= 4 Mi, wgk_gemv)
        self.kernel = "gemv_t_kernel" if trans else ("gemv_n_small_kernel" if R * C <= (4 << 20) and R >= 128 else "gemv_n_kernel")
        # graph_batch > 0: the dispatch is launch-bound (a few MB): record `graph_batch` dispatches into ONE command buffer
        # (a hipGraph) and replay it -- a step is then one Queue::submit of that buffer
        self.graph_batch = graph_batch
        self.launch_bound = R * C <= (4 << 20)

    def setup(self, wg, gpu, rank, world):
        self.wg, self.gpu, self.rank, self.world = wg, gpu, rank, world
        R, C = self.R, self.C
        self.m = device_random(wg, gpu, (R, C), self.np_dtype, 0xC000 + rank)
        vlen, olen = (R, C) if self.trans else (C, R)
        S = wg.BufferUsages
        if self.nrhs == 1:
            self.v = device_random(wg, gpu, (vlen,), self.np_dtype, 0xD000)
            self.out = wg.TensorBuilder.vector(olen, S.STORAGE | S.COPY_SRC).build(gpu.device(), self.np_dtype)
        else:
            self.v = device_random(wg, gpu, (vlen, self.nrhs), self.np_dtype, 0xD000)
            self.out = wg.TensorBuilder.matrix(olen, self.nrhs, S.STORAGE | S.COPY_SRC).build(gpu.device(), self.np_dtype)
        self.gemv = wg.Gemv.from_device(gpu.device())
        self.shapes = wg.ViewShapeBuffers()
        self.enc = gpu.device().create_command_encoder()
        self.pass_ = self.enc.compute_pass("bench", None)
        self.variant = wg.GemvVariant.GemvTr if self.trans else wg.GemvVariant.Gemv
        self.cmdbuf = None
        if self.graph_batch:
            self.step()  # sizes the split-K workspace before recording
            gpu.sync()
            enc = gpu.device().create_command_encoder(record=True)
            with enc.compute_pass("recorded", None) as p:
                for _ in range(self.graph_batch):
                    self.gemv.dispatch_generic(gpu.device(), self.shapes, p, self.out, self.m, self.v, self.variant)
            self.cmdbuf = enc.finish()

    def step(self):
        if self.cmdbuf is not None:
            self.gpu.queue().submit([self.cmdbuf])
        else:
            self.gemv.dispatch_generic(self.gpu.device(), self.shapes, self.pass_, self.out, self.m, self.v, self.variant)

    def _bytes(self):
        return np.dtype(self.np_dtype).itemsize * float(self.R * self.C + self.nrhs * (self.R + self.C))  # matrix + vector(s) + result(s)

    def units_per_step(self):
        return self._bytes() * self.world * max(self.graph_batch, 1)  # every rank streams its own matrix (no collective)

    def algorithmic_per_launch(self):
        return self._bytes()

    def launches_per_step(self):
        return max(self.graph_batch, 1)

    def check(self):
        gpu = self.gpu
        m = self.m.read(gpu.device()).reshape(self.R, self.C, order="F")
        vlen, olen = (self.R, self.C) if self.trans else (self.C, self.R)
        v = self.v.read(gpu.device()).astype(np.float64).reshape(vlen, self.nrhs, order="F")
        got = self.out.read(gpu.device()).astype(np.float64).reshape(olen, self.nrhs, order="F")
        idx = np.unique(np.random.default_rng(2).integers(0, olen, 64))
        a = (m[:, idx].T if self.trans else m[idx, :]).astype(np.float64)
        truth, sabs = a @ v, np.abs(a) @ np.abs(v)
        tol = 2 * np.sqrt(vlen) * 2.0 ** -24 * sabs + (2.0 ** -11 * np.abs(truth) + 2.0 ** -25 if self.dtype == "f16" else 0)
        assert (np.abs(got[idx] - truth) <= tol).all(), "bench sanity check failed (gemv)"
        scale = sabs if self.dtype == "f32" else np.abs(truth)  # (units: see GemmWorkload.check)
        ulp = np.spacing(scale.astype(self.np_dtype)).astype(np.float64)
        self.max_ulp_vs_f64 = float((np.abs(got[idx] - truth) / ulp).max())

    def cpu_baseline(self, budget_s):
        from oracle import wgsl_oracle as wo
        C = wo.CLib()
        R, Cc = self.R, self.C
        # BASELINE config 1 (1024 x 1024) and every matrix up to 64 MiB run IN FULL; the 1 GiB config-4 matrix on a 1/8
        # slice of its columns: same access pattern, bounded memory (128 MiB) and time
        scale = 1 if 4 * R * Cc <= (64 << 20) else 8
        Rs, Cs = R, Cc // scale
        m = rand_block(3, Rs * Cs, np.float32)
        vlen, olen = (Rs, Cs) if self.trans else (Cs, Rs)
        v, out = rand_block(4, vlen, np.float32), np.zeros(olen, np.float32)
        variant = wo.GEMV_TR if self.trans else wo.GEMV
        # work items of the port = workgroups of the reference kernel (gemv.wgsl: 64 invocations x 4 rows; gemv_tr: one output each)
        items = -(-olen // 256) if not self.trans else -(-olen // 64)
        dt, cores, reps = time_port(C, lambda: C.gemv(variant, out, wo.Shape(olen), m, wo.Shape(Rs, Cs), v, wo.Shape(vlen)), items, min(budget_s, 10.0))
        what = f"the {Rs}x{Cs} matrix in full" if scale == 1 else f"a {Rs}x{Cs} slice (1/{scale} of the matrix)"
        return {"value": 4.0 * (Rs * Cs + Rs + Cs) / dt / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
                "sample": f"oracle/wgsl_oracle.c `{'gemv_tr' if self.trans else 'gemv'}` on {what}, mean of {reps} runs on {cores} thread(s) "
                          f"({usable_cpus()} usable CPUs), {dt * 1e3:.3f} ms each" + (" (the f32 port: the reference has no f16 kernel)" if self.dtype == "f16" else "")}


class ReduceWorkload(Workload):
    bound = "hbm"
    metric = "reduce_gbs"
    unit = "GB/s"
    kernel = "reduce_rows4"

    def __init__(self, name, nvec, n):
        self.name, self.nvec, self.n = name, nvec, n

    def setup(self, wg, gpu, rank, world):
        self.wg, self.gpu, self.rank, self.world = wg, gpu, rank, world
        # 4096 vectors of 65536, each contiguous: a 65536 x 4096 column-major matrix
        self.x = device_random(wg, gpu, (self.n, self.nvec), np.float32, 0xE000 + rank)
        S = wg.BufferUsages
        self.res = wg.TensorBuilder.vector(self.nvec, S.STORAGE | S.COPY_SRC).build(gpu.device(), np.float32)
        self.red = wg.Reduce.new(gpu.device(), wg.ReduceOp.Sum)
        self.shapes = wg.ViewShapeBuffers()
        self.enc = gpu.device().create_command_encoder()
        self.pass_ = self.enc.compute_pass("bench", None)

    def step(self):
        self.red.dispatch_batched(self.gpu.device(), self.shapes, self.pass_, self.x, self.res)

    def _bytes(self):
        return 4.0 * (self.n * self.nvec + self.nvec)

    def units_per_step(self):
        return self._bytes() * self.world

    def algorithmic_per_launch(self):
        return self._bytes()

    def check(self):
        # (no oracle here: bit-exactness against the restated reduce.wgsl order is tests/test_gpu_parity.py's job; this is the bench's sanity check
        # against f64 within the re-association bound n 2^-24 sum|x| of SURVEY 8(c))
        gpu = self.gpu
        x = self.x.read(gpu.device())
        got = self.res.read(gpu.device())
        for c in (0, 1, self.nvec // 2, self.nvec - 1):
            col = x[c * self.n:(c + 1) * self.n].astype(np.float64)
            assert abs(float(got[c]) - col.sum()) <= self.n * 2.0 ** -24 * np.abs(col).sum(), "bench sanity check failed (reduce)"

    def cpu_baseline(self, budget_s):
        from oracle import wgsl_oracle as wo
        C = wo.CLib()
        nv = self.nvec // 8
        x = rand_block(5, self.n * nv, np.float32)
        dt, cores, reps = time_port(C, lambda: C.reduce_batched(wo.SUM, x, wo.Shape(self.n, nv)), nv, min(budget_s, 10.0))
        return {"value": 4.0 * (self.n * nv + nv) / dt / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
                "sample": f"oracle/wgsl_oracle.c reduce (128-lane order) on {nv} of the {self.nvec} vectors, one OpenMP task per vector, "
                          f"mean of {reps} runs on {cores} thread(s) ({usable_cpus()} usable CPUs), {dt * 1e3:.1f} ms each"}


class OpAssignWorkload(Workload):
    bound = "hbm"
    metric = "op_assign_gbs"
    unit = "GB/s"
    kernel = "op_assign_f32_vec"

    def __init__(self, name, n):
        self.name, self.n = name, n

    def setup(self, wg, gpu, rank, world):
        self.wg, self.gpu, self.rank, self.world = wg, gpu, rank, world
        self.a = device_random(wg, gpu, (self.n,), np.float32, 0xF000 + rank)
        self.b = device_random(wg, gpu, (self.n,), np.float32, 0xF100 + rank)
        self.op = wg.OpAssign.new(gpu.device(), wg.OpAssignVariant.Add)
        self.shapes = wg.ViewShapeBuffers()
        self.enc = gpu.device().create_command_encoder()
        self.pass_ = self.enc.compute_pass("bench", None)
        self.a0 = self.a.read(gpu.device())[:4096].copy()
        self.b0 = self.b.read(gpu.device())[:4096].copy()
        self.count = 0

    def step(self):
        self.op.dispatch(self.gpu.device(), self.shapes, self.pass_, self.a, self.b)
        self.count += 1

    def _bytes(self):
        return 12.0 * self.n  # read a, read b, write a

    def units_per_step(self):
        return self._bytes() * self.world

    def algorithmic_per_launch(self):
        return self._bytes()

    def check(self):
        exp = self.a0.copy()
        for _ in range(self.count):
            exp = exp + self.b0
        got = self.a.read(self.gpu.device())[:4096]
        assert got.tobytes() == exp.tobytes(), "bench sanity check failed (op_assign is not bit-exact)"

    def cpu_baseline(self, budget_s):
        from oracle import wgsl_oracle as wo
        C = wo.CLib()
        n = min(self.n, 1 << 26)
        a, b = rand_block(6, n, np.float32), rand_block(7, n, np.float32)
        dt, cores, reps = time_port(C, lambda: C.op_assign(wo.ADD, a, wo.Shape(n), b, wo.Shape(n)), n // 65536, min(budget_s, 5.0))
        return {"value": 12.0 * n / dt / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
                "sample": f"oracle/wgsl_oracle.c op_assign(Add) on {n} elements, mean of {reps} runs on {cores} thread(s) ({usable_cpus()} usable CPUs), {dt * 1e3:.1f} ms each"}


WORKLOADS = {
    "gemm_f32_4096": lambda: GemmWorkload("gemm_f32_4096", 4096, 4096, 4096, "f32"),
    "gemm_f16_8192": lambda: GemmWorkload("gemm_f16_8192", 8192, 8192, 8192, "f16"),
    "gemm_f16_32768": lambda: GemmWorkload("gemm_f16_32768", 32768, 32768, 32768, "f16"),
    # mid-size squares: fewer 256 x 256 tiles than CUs (f16: the 128 x 128 kernel; f32: planned split-K)
    "gemm_f16_2048": lambda: GemmWorkload("gemm_f16_2048", 2048, 2048, 2048, "f16"),
    "gemm_f32_2048": lambda: GemmWorkload("gemm_f32_2048", 2048, 2048, 2048, "f32"),
    # few columns (small batch): the streaming MFMA kernel, HBM-bound
    "gemm_f32_fewcols_32000x16x4096": lambda: FewColumnsGemmWorkload("gemm_f32_fewcols_32000x16x4096", 32000, 16, 4096),
    # tall-skinny (M >> N), the other shape family the north star names
    "gemm_f16_ts_131072x1024x8192": lambda: GemmWorkload("gemm_f16_ts_131072x1024x8192", 131072, 1024, 8192, "f16"),
    "gemm_f32_ts_65536x512x4096": lambda: GemmWorkload("gemm_f32_ts_65536x512x4096", 65536, 512, 4096, "f32"),
    "gemmtr_f16_8192": lambda: GemmWorkload("gemmtr_f16_8192", 8192, 8192, 8192, "f16", trans=True),
    "gemmtr_f16_32768": lambda: GemmWorkload("gemmtr_f16_32768", 32768, 32768, 32768, "f16", trans=True),
    # the same products on the reference's own operand distribution, U[0, 1) (gemm.rs:152 `new_random`); the clock of a power-capped part depends on it
    "gemm_f16_8192_u01": lambda: GemmWorkload("gemm_f16_8192_u01", 8192, 8192, 8192, "f16", values="u01"),
    "gemmtr_f16_8192_u01": lambda: GemmWorkload("gemmtr_f16_8192_u01", 8192, 8192, 8192, "f16", trans=True, values="u01"),
    "gemm_f16_32768_u01": lambda: GemmWorkload("gemm_f16_32768_u01", 32768, 32768, 32768, "f16", values="u01"),
    "gemm_f32_4096_u01": lambda: GemmWorkload("gemm_f32_4096_u01", 4096, 4096, 4096, "f32", values="u01"),
    # short K on many tiles: the per-tile prologue / pipeline drain / epilogue weigh most here (A/B shapes of tools/ab2.sh)
    "gemm_f16_8192x8192x512": lambda: GemmWorkload("gemm_f16_8192x8192x512", 8192, 8192, 512, "f16"),
    "gemm_f16_8192x8192x2048": lambda: GemmWorkload("gemm_f16_8192x8192x2048", 8192, 8192, 2048, "f16"),
    "gemm_f16_8192x8192x1024": lambda: GemmWorkload("gemm_f16_8192x8192x1024", 8192, 8192, 1024, "f16"),  # (the continuous tile walk's shapes: round 5)
    "gemmtr_f16_8192x8192x1024": lambda: GemmWorkload("gemmtr_f16_8192x8192x1024", 8192, 8192, 1024, "f16", trans=True),
    # 16 and 12 rounds of 256 tiles: where the dynamic tile scheduler starts to pay (WG_F16_SCHED=0|1 forces)
    "gemm_f16_16384x16384x8192": lambda: GemmWorkload("gemm_f16_16384x16384x8192", 16384, 16384, 8192, "f16"),
    "gemm_f16_16384x12288x8192": lambda: GemmWorkload("gemm_f16_16384x12288x8192", 16384, 12288, 8192, "f16"),
    "gemmtr_f32_4096": lambda: GemmWorkload("gemmtr_f32_4096", 4096, 4096, 4096, "f32", trans=True),
    # the ROW-major operator surface (wg_gemm_rm): GemmTr is the product with both operands contiguous along their output dimension (gemm_f16_nt.hip)
    "gemmtr_rm_f16_8192": lambda: RowMajorGemmWorkload("gemmtr_rm_f16_8192", 8192, 8192, 8192, "f16", trans=True),
    "gemm_rm_f16_8192": lambda: RowMajorGemmWorkload("gemm_rm_f16_8192", 8192, 8192, 8192, "f16"),
    "gemmtr_rm_f32_4096": lambda: RowMajorGemmWorkload("gemmtr_rm_f32_4096", 4096, 4096, 4096, "f32", trans=True),
    "gemv_f32_4096x65536": lambda: GemvWorkload("gemv_f32_4096x65536", 4096, 65536, False),
    "gemvtr_f32_65536x4096": lambda: GemvWorkload("gemvtr_f32_65536x4096", 65536, 4096, True),
    "gemv_f32_4096x65536_rhs8": lambda: GemvWorkload("gemv_f32_4096x65536_rhs8", 4096, 65536, False, nrhs=8),
    "gemv_f16_4096x65536": lambda: GemvWorkload("gemv_f16_4096x65536", 4096, 65536, False, dtype="f16"),
    "gemvtr_f16_65536x4096": lambda: GemvWorkload("gemvtr_f16_65536x4096", 65536, 4096, True, dtype="f16"),
    "gemv_f32_1024": lambda: GemvWorkload("gemv_f32_1024", 1024, 1024, False),
    "gemv_f32_1024_graph": lambda: GemvWorkload("gemv_f32_1024_graph", 1024, 1024, False, graph_batch=64),
    "reduce_f32_4096x65536": lambda: ReduceWorkload("reduce_f32_4096x65536", 4096, 65536),
    "op_assign_f32_256M": lambda: OpAssignWorkload("op_assign_f32_256M", 1 << 28),
}
# Headline: the north-star's M-sharded f16 GEMM (BASELINE configs[4]); the SAME problem at every --gpus N ("strong"), so the
# driver's per-N values are comparable. It fits one GPU (3 x 2 GiB), which makes it the N = 1 workload as well.
DEFAULT_WORKLOAD = "gemm_f16_32768"
# (Config 1's two workloads run FIRST, right behind the headline: the replayed dispatch's time depends on what the process allocated before it -- 2.6 us per dispatch in a fresh
# process, 3.4-5.7 us behind 25 other workloads on the same chip, profiles/r06_c1_replay_populations.txt -- and the figure wanted here is the kernel's, not the allocator's.)
SECONDARY = ["gemv_f32_1024", "gemv_f32_1024_graph", "gemm_f16_8192", "gemmtr_f16_8192", "gemm_f16_8192x8192x1024", "gemmtr_f16_8192x8192x1024", "gemmtr_f16_32768", "gemm_f16_32768_u01", "gemm_f16_8192_u01", "gemmtr_f16_8192_u01", "gemm_f32_4096", "gemm_f32_4096_u01", "gemm_f16_2048", "gemm_f32_2048", "gemm_f16_ts_131072x1024x8192", "gemm_f32_ts_65536x512x4096", "gemm_f32_fewcols_32000x16x4096", "gemmtr_rm_f16_8192", "gemv_f32_4096x65536", "gemvtr_f32_65536x4096", "gemv_f32_4096x65536_rhs8", "gemv_f16_4096x65536", "gemvtr_f16_65536x4096", "reduce_f32_4096x65536",
             "op_assign_f32_256M"]


# With N > 1 ranks the HBM-bound operators shard by independent units with no data-path collective: every rank
# streams its own row block of an N-times taller matrix ("weak"); (name, fixed step count -- the same on every rank).
DIST_SECONDARY = [("gemv_f32_4096x65536", 2000), ("gemvtr_f32_65536x4096", 2000), ("reduce_f32_4096x65536", 2000)]


def _pmc_row(workload: str):
    """The workload's row of the tracked counter summary: profiles/r06_pmc.csv (tools/pmc.sh: ONE rocprofv3 run per workload, separate
    --pmc passes, --kernel-trace only), else the previous round's. Static, measured on the builder's box -- every field taken from it
    carries `_profiled` in its name (or says so in `traffic_source`)."""
    import csv
    for fn in ("r06_pmc.csv", "r05_pmc.csv", "r04_pmc.csv", "r03_pmc.csv", "r02_pmc.csv"):
        path = os.path.join(ROOT, "profiles", fn)
        try:
            with open(path) as f:
                for r in csv.DictReader(ln for ln in f if not ln.startswith("#")):
                    if r["workload"] == workload:
                        return r, fn
        except Exception:
            continue
    return None, None


def load_traffic(workload: str):
    """HBM-side bytes per launch of the dominant kernel from the PER-WORKLOAD counter pass (FETCH_SIZE doubled per the guide's gfx950
    correction + WRITE_SIZE): (bytes, source file) or (None, None)."""
    r, fn = _pmc_row(workload)
    try:
        return (int(float(r["traffic_bytes"])), fn) if r and r.get("traffic_bytes", "") != "" else (None, None)
    except Exception:
        return None, None


def load_pmc(workload: str):
    """Counter-derived facts of the workload's dominant kernel from the same row: MFMA utilisation in cycles, L2 hit rate and the effective
    clock of the PROFILED run (profiled passes clock lower than unprofiled ones)."""
    r, _ = _pmc_row(workload)
    out = {}
    if r:
        for k in ("mfma_util", "clock_ghz", "l2_hit_rate", "lds_bank_conflict"):
            if r.get(k, "") != "":
                try:
                    out[k] = float(r[k])
                except ValueError:
                    pass
    return out


def run_workload(wg, gpu, name, steps, warmup, rank, world, barrier, with_cpu, cpu_budget, min_seconds=0.0, keep=True, agree=None):
    """Times EXACTLY `steps` steps (after `warmup` untimed ones). With min_seconds > 0 (secondary configs only) `steps` is raised
    so that the timed region lasts at least that long: millisecond kernels timed for a few tens of ms still see the clocks
    ramping (measured: f32 GEMM 140 TF over 30 ms vs 148 TF sustained)."""
    w = WORKLOADS[name]()
    setup_err = None
    try:
        w.setup(wg, gpu, rank, world)
    except Exception as e:
        if agree is None:
            raise
        setup_err = e
    if agree is not None:
        # multi-rank: a rank whose set-up failed (no peer mapping, out of memory, ...) must not leave the others waiting inside a
        # collective or on its flags -- everybody learns about it here and skips the run together
        if not agree(setup_err is None):
            raise RuntimeError(f"set-up failed on {'this' if setup_err is not None else 'another'} rank" + (f": {type(setup_err).__name__}: {setup_err}" if setup_err else ""))
    t_w = time.perf_counter()
    for _ in range(max(warmup, 1)):
        w.step()
    if hasattr(w, "finish"):
        w.finish()
    gpu.sync()
    if min_seconds > 0:
        per_step = max((time.perf_counter() - t_w) / max(warmup, 1), 1e-6)
        steps = int(min(max(steps, min_seconds / per_step), 20000))
        for _ in range(min(steps, 50)):  # a little more warm-up at the final cadence
            w.step()
    ts = wg.GpuTimestamps.new(gpu.device(), 2)
    barrier()
    gpu.sync()
    clock = gpu.clock_probe()  # one stamp kernel ahead of the timed steps (s_memtime against s_memrealtime per XCD), its twin after them
    t0 = time.perf_counter()
    ts.write(gpu.device())
    for _ in range(steps):
        w.step()
    if hasattr(w, "finish"):
        w.finish()  # (pipelined multi-rank steps: the last step's deferred panel; no-op otherwise)
    ts.write(gpu.device())
    # the closing stamp goes in right behind the last step, in stream order, BEFORE the stream drains and before the inter-rank barrier: the interval
    # must not include idle time at the idle clock (a few microseconds of stamp kernel inside the timed region against a biased clock otherwise)
    clock_ghz = clock.end()  # {"mean", "min", "max"} over the XCDs: the shader clock the timed steps actually ran at (None if unavailable); synchronises
    gpu.sync()
    own_elapsed = time.perf_counter() - t0  # this rank's own steps, drained, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    ev = ts.wait_for_results_ms()
    launches = getattr(w, "launches_per_step", lambda: 1)()
    kernel_ms = (ev[1] - ev[0]) / (steps * launches)  # HIP events on the stream the kernels run on
    if not os.environ.get("WG_BENCH_NO_CHECK"):
        w.check()
    if agree is not None:
        barrier()  # every rank is past its checks: nobody reads a peer's buffers any more
        if hasattr(w, "close") and not keep:
            w.close()
        barrier()  # ... and nobody frees a buffer a peer still has mapped for copying
    res = {"workload": w if keep else None, "elapsed": elapsed, "own_elapsed": own_elapsed, "kernel_ms": kernel_ms, "steps": steps, "clock": clock_ghz,
           "max_ulp_vs_f64": getattr(w, "max_ulp_vs_f64", None)}
    res["cpu"] = w.cpu_baseline(cpu_budget) if (with_cpu and rank == 0 and world == 1) else None
    return res


MFMA_CEILING = None  # {"tflops", "clock_ghz"}: wg_debug_mfma_ceiling, measured once per run at N = 1 (main)


def summarize(w, elapsed, kernel_ms, steps, world, clock=None):
    scale = 1e12 if w.unit == "TFLOP/s" else 1e9
    value = w.units_per_step() * steps / elapsed / scale
    achieved = w.algorithmic_per_launch() / (kernel_ms * 1e-3) / scale
    peak = PEAK_MFMA_TFLOPS[w.dtype] if w.bound == "mfma" else PEAK_HBM_GBS
    # the PMC traffic figure was measured on the single-GPU, one-launch-per-step form of the workload: null for any other launch shape
    launches = getattr(w, "launches_per_step", lambda: 1)()
    traffic, traffic_src = load_traffic(w.name) if (world == 1 and launches in (1, getattr(w, "graph_batch", 0))) else (None, None)
    roof = {"bound": w.bound, "kernel": w.kernel, "achieved": round(achieved, 3), "peak": peak, "unit": w.unit,
            "frac": round(achieved / peak, 4), "kernel_ms": round(kernel_ms, 5), "traffic": traffic}
    if traffic is not None:
        roof["traffic_source"] = f"profiles/{traffic_src}: tracked per-workload rocprofv3 --pmc pass on the builder's box, not this run"
        roof["algorithmic_bytes"] = int(w.algorithmic_bytes())
    if world == 1:
        pmc = load_pmc(w.name)  # static counters of the tracked profile: every one of them is named *_profiled
        if w.bound == "mfma" and "mfma_util" in pmc:
            roof["mfma_util_profiled"] = pmc["mfma_util"]  # SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x cycles)
        if "clock_ghz" in pmc:
            roof["clock_ghz_profiled"] = pmc["clock_ghz"]
        if "l2_hit_rate" in pmc:
            roof["l2_hit_rate_profiled"] = pmc["l2_hit_rate"]
    if clock:
        # measured IN THIS RUN around the timed steps: mean shader clock (s_memtime / s_memrealtime, mean over the XCDs). For an MFMA-bound kernel
        # `frac` = (issue efficiency) x (clock / 2.4 GHz): a slow box and a slow kernel look different here
        roof["clock_ghz_measured"] = round(clock["mean"], 3)
        roof["clock_ghz_xcd_min_max"] = [round(clock["min"], 3), round(clock["max"], 3)]
        if w.bound == "mfma":
            roof["frac_at_measured_clock"] = round(achieved / (peak * clock["mean"] / 2.4), 4)  # share of the matrix cores' rate at the clock they got
    if w.bound == "mfma" and w.dtype == "f16" and MFMA_CEILING:
        # what the package power cap leaves the matrix cores ALONE on random operands (v_mfma only, in-register operands), measured in this run
        roof["mfma_only_ceiling_tflops"] = round(MFMA_CEILING["tflops"], 1)
        roof["mfma_only_ceiling_clock_ghz"] = round(MFMA_CEILING["clock_ghz"], 3)
        roof["frac_of_ceiling"] = round(achieved / MFMA_CEILING["tflops"], 4)
    if getattr(w, "launch_bound", False):
        roof["dispatch_us"] = round(kernel_ms * 1e3, 2)  # launch-bound: wall time per dispatch on the stream, eager or replayed
    return value, roof


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (one process per GPU) as CHILDREN of this process --
    which has not touched the GPU and never will (no exec of a GPU-initialised process) -- relay rank 0's single JSON line,
    and fail if any rank fails."""
    import subprocess
    if not args.dry_run:
        import torch  # device_count() does not initialise the GPU on this image
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and os.environ.get("WG_BENCH_OVERSUBSCRIBE") != "1":
            log(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible on this node; one process per GPU is required "
                f"(WG_BENCH_OVERSUBSCRIBE=1 shares GPUs between ranks for plumbing tests: the staged copy-engine exchange only, RCCL refuses it)")
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL, hipIpcGetMemHandle)
    # HIP multiplexes a process's streams onto 4 hardware queues by default; a rank has its compute stream, the collective's stream and one
    # copy stream per peer: give every stream its own queue, or copies queue up behind Gemms instead of running beside them
    env.setdefault("GPU_MAX_HW_QUEUES", "16")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] launching:", " ".join(cmd))
    # its own process group and a time limit: a rank that hangs (a collective that never completes on this node) must not hold the line
    # until somebody else's clock runs out. (The exchange-engine trials inside the ranks have their own, shorter limits: engine_trials.)
    limit = float(os.environ.get("WG_BENCH_LAUNCH_TIMEOUT", "1700"))
    rc, out = run_group(cmd, env, limit)
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and ln.rstrip().endswith("}")]
    if rc != 0 or not lines:
        sys.stderr.write(out)
        log(f"[bench] the {args.gpus}-rank run failed ({'timed out after %.0f s' % limit if rc == -9 else 'rc %d' % rc})")
        return rc if rc > 0 else 1
    print(lines[-1], flush=True)
    return 0


def run_group(cmd, env, limit_s):
    """Run `cmd` as the leader of a NEW process group, stdout captured (stderr passes through); on expiry of `limit_s` kill the whole group
    (exactly the processes started here and their descendants -- never by pattern). Returns (returncode or -9 on time-out, stdout)."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=limit_s)
        return p.returncode, out
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        try:
            out, _ = p.communicate(timeout=30)
        except Exception:
            out = ""
        return -9, out or ""


# The exchange engines a multi-rank run tries (--gather auto). "rccl_cus4" is the RCCL engine with 4 instead of 8 CUs left to RCCL's copy kernels
# (profiles/r03_evidence.md section 12: which of the two is faster can only be measured where RCCL has peers -- so the run measures it).
ENGINES = ["rccl", "rccl_cus4", "staged"]
ENGINE_COMM_CUS = {"rccl": 8, "rccl_cus4": 4}  # CUs the Gemm's stream leaves free, all from one XCD (wg_ctx_create_with_cu_count_one_xcd)
_TRIAL_STORE = None


def engine_trials(args, rank, world):
    """--gather auto with N > 1 ranks: every exchange engine gets a short trial -- as a FRESH child group (this rank starts one child per
    engine, all ranks' children of one engine rendezvous on their own port), with a time limit, BEFORE this process has touched the GPU
    (it never re-executes itself and starts children only while it is still GPU-free). An engine that fails is recorded with its
    error, one that HANGS (RCCL across ranks has never run on some nodes) with "timeout" after WG_BENCH_TRIAL_TIMEOUT seconds (default
    240): its group is killed and the next engine is tried. Returns {engine: {"ms_per_step": float | None, "error": str | None}} as seen by
    THIS rank; the ranks agree on the winner afterwards (max over ranks, failures = infinity)."""
    limit = float(os.environ.get("WG_BENCH_TRIAL_TIMEOUT", "240"))
    base_port = int(os.environ.get("MASTER_PORT", "29500"))
    argv = [a for a in sys.argv[1:]]
    out = {}
    # The ranks start every trial TOGETHER (a key-value store of their own, no process group, no GPU): an engine that fails at once on one rank
    # and hangs on another would otherwise leave the first rank's next trial waiting for the second rank's child for most of ITS time limit --
    # and the next engine would be written off as "timeout" for no fault of its own.
    import datetime
    from torch.distributed import TCPStore
    global _TRIAL_STORE
    try:
        store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), base_port + 40, world, rank == 0, timeout=datetime.timedelta(seconds=120))
        _TRIAL_STORE = store  # rank 0 hosts it: it must outlive every other rank's last poll, i.e. live as long as this process
    except Exception as e:  # noqa: BLE001 -- (port taken, ...): the trials still run, each rank on its own clock as before
        store = None
        log(f"[bench] rank {rank}: no store for the trials ({type(e).__name__}: {e}); trials start unsynchronised")

    def together(tag):
        if store is None:
            return
        store.add(tag, 1)
        t_end = time.time() + limit + 120
        while store.add(tag, 0) < world:  # (add 0 reads the counter)
            if time.time() > t_end:
                raise SystemExit(f"[bench] rank {rank}: the other ranks never reached engine trial {tag}")
            time.sleep(0.05)

    for k, mode in enumerate(ENGINES):
        together(f"trial{k}")
        env = dict(os.environ)
        env["MASTER_PORT"] = str(base_port + 1 + k)       # this engine's own rendezvous, hosted by rank 0's child
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)       # (the launcher agent's store only serves the original port)
        env["WG_BENCH_TRIAL"] = mode
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--gather", mode]
        t0 = time.perf_counter()
        rc, text = run_group(cmd, env, limit)
        res = None
        for ln in text.splitlines():
            if ln.startswith("{") and '"trial"' in ln:
                try:
                    res = json.loads(ln)
                except ValueError:
                    pass
        if rc == -9:
            out[mode] = {"ms_per_step": None, "error": "timeout"}
        elif rc != 0 or res is None or res.get("ms_per_step") is None:
            out[mode] = {"ms_per_step": None, "error": (res or {}).get("error") or f"trial exited with rc {rc}"}
        else:
            out[mode] = {"ms_per_step": float(res["ms_per_step"]), "error": None}
        log(f"[bench] rank {rank}: engine trial {mode}: {out[mode]} ({time.perf_counter() - t0:.1f} s)")
    return out


def dry_run(args, rank, world, trials=None, trial_mode=None) -> None:
    """--dry-run: the launcher / rendezvous / max-over-ranks / one-JSON-line plumbing of a multi-rank run WITHOUT a GPU (gloo, host
    arithmetic): the M-shard planner and the pipelined all-gather driver (wgmath_amd/sharded.py) with a NumPy GEMM standing in for
    the HIP kernel. Test infrastructure for tests/test_bench_launcher.py; its line says so and carries no roofline."""
    import torch
    import torch.distributed as dist
    from wgmath_amd.sharded import MShardPlan, ShardedGemm  # loads the built library (no fallback), touches no GPU
    if trial_mode:
        # a trial child of the dry run: WG_BENCH_DRY_HANG / WG_BENCH_DRY_FAIL = "<engine>:<rank>" make that rank of that engine's trial hang
        # forever / raise (tests/test_bench_launcher.py: what a collective that never completes on a real node looks like to the launcher)
        for var, act in (("WG_BENCH_DRY_HANG", "hang"), ("WG_BENCH_DRY_FAIL", "fail")):
            for spec in filter(None, os.environ.get(var, "").split(",")):
                m, r = spec.split(":")
                if m == trial_mode and int(r) == rank:
                    if act == "hang":
                        time.sleep(10 ** 6)
                    raise RuntimeError(f"dry-run: engine {m} made to fail on rank {r}")
    dist.init_process_group("gloo")
    chosen, report = None, None
    if trials is not None:
        chosen, report = agree_on_engine(trials, lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX), "cpu")
    M, N, K = 64 * world, 96, 32
    pl = MShardPlan(M, N, K, world, npanels=3)
    rng = np.random.default_rng(7)
    A, B = rng.standard_normal((M, K)).astype(np.float32), rng.standard_normal((K, N)).astype(np.float32)
    r0, nr = pl.a_rows(rank)
    a_g, b_flat = np.ascontiguousarray(A[r0:r0 + nr].reshape(-1, order="F")), B.reshape(-1, order="F")
    gathered = torch.zeros(pl.gathered_elems(), dtype=torch.float32)
    g_np = gathered.numpy()

    def idx(sh):
        return sh.offset + np.arange(sh.size[0])[:, None] + np.arange(sh.size[1])[None, :] * sh.stride

    def local_gemm(o, a, b):
        g_np[idx(o)] = a_g[idx(a)] @ b_flat[idx(b)]

    def all_gather(start, count, rk):
        outs = [gathered[start + g * count:start + (g + 1) * count] for g in range(world)]
        return dist.all_gather(outs, gathered[start + rk * count:start + (rk + 1) * count].clone(), async_op=True)

    waits = []  # (panel, ms) of every wait for a panel's all-gather, the dry run's stand-in for wg_comm_wait_times

    def timed_wait(w):
        t = time.perf_counter()
        w.wait()
        waits.append((len(waits) % pl.npanels, (time.perf_counter() - t) * 1e3))

    drv = ShardedGemm(pl, rank, local_gemm, all_gather, wait=timed_wait)
    for _ in range(args.warmup):
        drv.step()
    del waits[:]
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        drv.step()
    own = time.perf_counter() - t0
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    C = A @ B
    ok = all(abs(g_np[pl.element_index(r, c)] - C[r, c]) < 1e-3 for r in range(0, M, 7) for c in range(0, N, 5))
    if not ok:
        raise SystemExit("dry-run: gathered product is wrong")
    if trial_mode:
        print(json.dumps({"trial": trial_mode, "ms_per_step": round(float(el.item()) / args.steps * 1e3, 5)}, allow_nan=False), flush=True)
        dist.destroy_process_group()
        return
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {"ms_per_step": own / max(args.steps, 1) * 1e3, "clock_ghz": None, "waits": waits})
    if rank == 0:
        elapsed = float(el.item())
        extra = {} if report is None else dict(report, gather_engine=chosen)
        detail, flat = ranks_detail(per_rank, pl.npanels)
        extra.update(flat)
        emit({"metric": "dry_run_gemm_tflops", "value": round(2.0 * M * N * K * args.steps / elapsed / 1e12, 9), "unit": "TFLOP/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                          "data": "dry-run: host arithmetic over gloo, no GPU (launcher plumbing test)",
                          "config": {"workload": f"dry_run_{M}x{N}x{K}", "ranks": dist.get_world_size(),
                                     "parallelism": f"m-shard x{world} + gloo all-gather (dry run)",
                                     "all_gather_bytes_per_step": (world - 1) * pl.Mg * N * 4,
                                     "comm_compute_units": ENGINE_COMM_CUS.get(chosen or args.gather, 0), "rccl_reported_ranks": 0, **extra},
                          "ranks_detail": detail}, args.detail, lambda text: print(text, flush=True))
    dist.destroy_process_group()


def ranks_detail(per_rank, npanels):
    """What tells a slow link from a slow rank from a slow clock in ONE multi-rank run. `per_rank`: one dict per rank -- {"ms_per_step": its own
    timed region / steps, "clock_ghz": its measured shader clock (None if unavailable), "waits": [(panel, ms the compute stream stood still for that
    panel's exchange), ...] from a few extra steps after the timed region (wg_comm_set_wait_timing)}. Returns (detail object, flat scalars for `config`).
    Reading it: one rank's ms_per_step above the others with a lower clock = a slow chip; every rank waiting on the same early panels = the links
    (or RCCL's share of CUs) do not keep up with the Gemm; waits on the LAST panel only = the un-hidden tail, what a taper / pipelined steps shrink."""
    ms = [float(r["ms_per_step"]) for r in per_rank]
    clk = [r.get("clock_ghz") for r in per_rank]
    wait = np.zeros((len(per_rank), max(1, npanels)))
    cnt = np.zeros_like(wait)
    for i, r in enumerate(per_rank):
        for p, t in r.get("waits") or []:
            if 0 <= int(p) < wait.shape[1]:
                wait[i, int(p)] += float(t)
                cnt[i, int(p)] += 1
    mean = np.where(cnt > 0, wait / np.maximum(cnt, 1), 0.0)          # per rank, per panel: mean wait of a step
    per_step_total = mean.sum(axis=1)
    detail = {"ms_per_step": [round(x, 4) for x in ms], "clock_ghz": clk,
              "panel_wait_ms_max_over_ranks": [round(float(x), 4) for x in mean.max(axis=0)],
              "panel_wait_rank_of_max": [int(x) for x in mean.argmax(axis=0)],
              "wait_ms_per_step": [round(float(x), 4) for x in per_step_total]}
    known = [c for c in clk if c is not None]
    flat = {"rank_ms_per_step_min": round(min(ms), 4), "rank_ms_per_step_max": round(max(ms), 4), "rank_slowest": int(np.argmax(ms)),
            "rank_clock_ghz_min": round(min(known), 3) if known else None, "rank_clock_ghz_max": round(max(known), 3) if known else None,
            "exchange_wait_ms_per_step_max": round(float(per_step_total.max()), 4),
            "exchange_wait_ms_last_panel_max": round(float(mean[:, -1].max()), 4),
            "exchange_wait_ms_before_last_panel_max": round(float(mean[:, :-1].sum(axis=1).max()) if mean.shape[1] > 1 else 0.0, 4)}
    return detail, flat


def agree_on_engine(trials, all_reduce_max, device):
    """Every rank ran its own child of every engine's trial: the ranks agree on max-over-ranks times (a failure anywhere = infinity) and
    pick the fastest engine that worked everywhere. Returns (engine, {"engine_trials": ..., "chosen": ...}); exits if none works."""
    import torch
    t = torch.tensor([trials[m]["ms_per_step"] if trials.get(m, {}).get("ms_per_step") is not None else float("inf") for m in ENGINES],
                     dtype=torch.float64, device=device)
    all_reduce_max(t)
    agreed = {m: float(v) for m, v in zip(ENGINES, t.tolist())}
    rep = {m: {"ms_per_step": round(agreed[m], 4) if np.isfinite(agreed[m]) else None,
               "error": None if np.isfinite(agreed[m]) else (trials.get(m, {}).get("error") or "failed on another rank")} for m in ENGINES}
    ok = {m: v for m, v in agreed.items() if np.isfinite(v)}
    if not ok:
        sys.exit(f"no exchange engine works on this node: {rep}")
    best = min(ok, key=ok.get)
    return best, {"engine_trials": rep, "chosen": best}


def main():
    global VALUES
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--dist", default=VALUES, choices=["pm1", "u01", "zero"],
                    help="operand distribution of the run: pm1 = U[-1,1) (default, the headline), u01 = U[0,1) (the reference's new_random); the *_u01 "
                         "secondary workloads always use u01")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra single-GPU configs (reported under `others` in the --detail sidecar, as scalars under `targets` in the line)")
    ap.add_argument("--skip", default="", help="comma-separated secondary workloads to skip (e.g. under rocprofv3)")
    ap.add_argument("--secondary-seconds", type=float, default=0.6, help="minimum timed duration of each secondary config")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU work for the cpu_baseline sample")
    ap.add_argument("--gather", default=os.environ.get("WG_BENCH_GATHER", "auto"), choices=["auto"] + ENGINES,
                    help="exchange engine of the M-sharded Gemm with N > 1 ranks: RCCL all-gather (+ relayout) with 8 or 4 CUs left to its kernels, "
                         "staged contiguous peer copies (+ relayout), or auto = a short trial of each as a fresh child group, the timed steps on the fastest")
    ap.add_argument("--dry-run", action="store_true", help="multi-rank plumbing test without a GPU (gloo, host arithmetic)")
    ap.add_argument("--detail", default=os.environ.get("WG_BENCH_DETAIL", DEFAULT_DETAIL),
                    help="sidecar file for the full record (the stdout line + `others`: every secondary workload's roofline / cpu_baseline objects, and "
                         "`ranks_detail`); '-' = none. The same text goes to stderr as `[bench detail] {...}`")
    args = ap.parse_args()
    VALUES = args.dist
    os.environ["WG_BENCH_VALUES"] = args.dist  # (self-launched ranks inherit it)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))  # before anything touches the GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    trial_mode = os.environ.get("WG_BENCH_TRIAL")  # set: this process is one rank of ONE engine's trial (a child of engine_trials)
    trials = None
    if world > 1 and args.gather == "auto" and not trial_mode:
        # (importing torch initialises no GPU; done HERE so that the first trial's clock does not run while a fresh box pages the image's
        # libraries in -- 1-2 minutes the first time -- which would time the first engine out for no fault of its own)
        import torch  # noqa: F401
        trials = engine_trials(args, rank, world)  # fresh child groups with time limits, while this process is still GPU-free
    if args.dry_run:
        return dry_run(args, rank, world, trials, trial_mode)

    global DIST
    if world > 1:
        # before the HIP runtime initialises: one hardware queue per stream (compute, collective, one copy stream per peer), or the copies of
        # the exchange queue up behind the Gemms instead of running beside them (HIP's default folds all streams onto 4 queues)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    # WG_BENCH_FORCE_DIST=1: take the communicator path even with one rank (single-GPU test of the N > 1 plumbing: RCCL with one rank)
    dist_mode = world > 1 or os.environ.get("WG_BENCH_FORCE_DIST") == "1"
    if dist_mode:
        # torch bundles its own ROCm runtime (same soname as /opt/rocm's): it must be loaded FIRST so that libwgebra_hip.so
        # binds to that one copy -- two HIP runtimes in one process cannot both drive the GPU (wgmath_amd checks for this)
        import torch  # noqa: F401
    import wgmath_amd as wg

    oversub = False
    saved_stdout = None
    if dist_mode:
        # stdout carries exactly ONE line (the JSON): RCCL prints a version banner to stdout when its first communicator is created --
        # send everything the libraries print during the run to stderr, and give stdout back just before the line is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch
        import torch.distributed as dist
        from wgmath_amd.sharded import Comm, new_unique_id
        # (after a trial that had to be killed the driver may still be resetting the device: "No HIP GPUs are available" for a while -- retry)
        gpu_deadline = time.time() + (120.0 if trials and any(v.get("error") == "timeout" for v in trials.values()) else 0.0)
        while True:
            try:
                ndev = torch.cuda.device_count()
                oversub = world > ndev  # WG_BENCH_OVERSUBSCRIBE: ranks share GPUs -- RCCL refuses that, so gloo is the control plane and the
                dev_index = local_rank % max(ndev, 1)  # data plane is peer copies between the ranks' buffers on the shared device
                torch.cuda.set_device(dev_index)
                break
            except RuntimeError as e:
                if time.time() >= gpu_deadline:
                    raise
                log(f"[bench] rank {rank}: GPU not available yet after a killed trial ({e}); retrying")
                time.sleep(5.0)
        # torch.distributed is the CONTROL plane only (launch contract: rendezvous, barrier, max-over-ranks, shipping the unique id and
        # the IPC handles); the data plane -- RCCL all-gather or SDMA peer copies -- is driven by libwgebra_hip.so itself (wg_comm_*)
        if oversub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_index}"))
        total_cus = torch.cuda.get_device_properties(dev_index).multi_processor_count
        gather = args.gather if not oversub else "staged"  # (ranks sharing a GPU: RCCL refuses that)
        # CU partitioning between compute and communication. Every f16 GEMM workgroup needs a whole CU (160 KiB
        # LDS, 512 registers per lane), so RCCL's copy kernels, launched from a second queue while a GEMM grid is resident, only get CUs
        # when GEMM workgroups retire (tools/overlap_probe.py: they start ~0.75-1.4 ms late). For the RCCL engine the GEMM stream is
        # therefore CU-masked to leave `comm_cus` CUs free. Round 2 needed 32 of them (a mask had to take whole shader engines for the
        # static tile map); the rank's product is ONE scheduler-driven launch per step now (wg_comm_set_one_launch) and 8 -- one CU per
        # XCD -- are enough: 248 CUs compute. The copy engines of the staged / peer engines need none: all 256 CUs compute.
        def make_gpu(mode):
            # WG_BENCH_COMM_CUS overrides the engine's own figure (experiments)
            comm_cus = int(os.environ.get("WG_BENCH_COMM_CUS", ENGINE_COMM_CUS.get(mode, 0))) if (mode.startswith("rccl") and world > 1) else 0
            if comm_cus > 0 and total_cus > 2 * comm_cus:
                try:
                    return wg.GpuInstance.new(dev_index, cu_count=total_cus - comm_cus, one_xcd=os.environ.get("WG_BENCH_CU_MASK_SPREAD") != "1")
                except Exception as e:  # no CU-masked stream on this runtime
                    log(f"[bench] CU-masked compute stream unavailable ({e}); using an unmasked stream")
            return wg.GpuInstance.new(dev_index)

        def bcast_id():
            box = [new_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        def all_gather_object(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out

        def barrier():
            gpu.sync()
            dist.barrier()
            torch.cuda.synchronize()

        def agree(ok: bool) -> bool:  # True iff every rank says ok
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if oversub else f"cuda:{dev_index}")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        trial_report = None
        if trials is not None:  # the engines' trials ran as child groups (engine_trials): agree on the winner, build only that engine
            gather, trial_report = agree_on_engine(trials, lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX), "cpu" if oversub else f"cuda:{dev_index}")
            if oversub:
                gather = "staged"  # (ranks sharing a GPU: RCCL refuses that; its trials ran as the staged engine too)
        engines = {}  # mode -> (GpuInstance, Comm)
        for mode in (ENGINES if gather == "auto" else [gather]):
            g = make_gpu(mode)
            engines[mode] = (g, Comm(g, world, rank, None if oversub else bcast_id()))
        first = next(iter(engines))
        gpu, comm = engines[first]
        DIST = {"comm": comm, "mode": first, "barrier": lambda: dist.barrier(), "all_gather_object": all_gather_object}
    else:
        gpu = wg.GpuInstance.new(local_rank)

        def barrier():
            gpu.sync()

    info = gpu.adapter()
    _pad = None
    if os.environ.get("WG_BENCH_PAD"):  # experiment hook: shift every later allocation by this many bytes
        _pad = wg.TensorBuilder.vector(int(os.environ["WG_BENCH_PAD"]) // 4, wg.BufferUsages.STORAGE).build(gpu.device(), np.float32)

    dist_report = trial_report if dist_mode else None
    if dist_mode and trial_mode:
        # one rank of one engine's trial: a few untimed-contract steps, the max over ranks on stdout, done
        import torch
        import torch.distributed as dist
        try:
            r = run_workload(wg, gpu, args.workload, max(2, args.warmup), 1, rank, world, barrier, False, 0.0, keep=False, agree=agree)
            el, err = r["elapsed"] / r["steps"], None
        except Exception as e:
            el, err = float("inf"), f"{type(e).__name__}: {e}"
        t = torch.tensor([el], dtype=torch.float64, device=f"cuda:{dev_index}" if not oversub else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps({"trial": trial_mode, "ms_per_step": round(float(t.item()) * 1e3, 4) if np.isfinite(float(t.item())) else None, "error": err}, allow_nan=False), flush=True)
        os.dup2(2, 1)
        for g, cm in engines.values():
            cm.close()
        dist.destroy_process_group()
        sys.exit(0 if err is None and np.isfinite(float(t.item())) else 1)
    if dist_mode and len(engines) > 1:
        # auto: a short untimed trial of both engines (warm-up steps each, max over ranks), then the contract's timed run on the faster;
        # an engine that fails on this node (e.g. no peer mapping) is reported and skipped, never silently replaced
        import torch
        import torch.distributed as dist
        trial = {}
        for mode, (g, cm) in engines.items():
            DIST.update(comm=cm, mode=mode)
            gpu = g
            try:
                r = run_workload(wg, g, args.workload, max(2, args.warmup), 1, rank, world, barrier, False, 0.0, keep=False, agree=agree)
                el = r["elapsed"] / r["steps"]
                err = None
            except Exception as e:
                el, err = float("inf"), f"{type(e).__name__}: {e}"
            t = torch.tensor([el], dtype=torch.float64, device=f"cuda:{dev_index}" if not oversub else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            trial[mode] = {"ms_per_step": None if not np.isfinite(float(t.item())) else round(float(t.item()) * 1e3, 4), "error": err}
        ok = {m: v["ms_per_step"] for m, v in trial.items() if v["ms_per_step"] is not None}
        if not ok:
            sys.exit(f"no exchange engine works on this node: {trial}")
        best = min(ok, key=ok.get)
        gpu, comm = engines[best]
        DIST.update(comm=comm, mode=best)
        dist_report = {"engine_trials": trial, "chosen": best}
        info = gpu.adapter()

    main_res = run_workload(wg, gpu, args.workload, args.steps, args.warmup, rank, world, barrier,
                            not args.no_cpu_baseline, args.cpu_budget, agree=agree if dist_mode else None)
    elapsed = main_res["elapsed"]
    rank_detail, rank_flat = None, {}
    if dist_mode:
        import torch
        import torch.distributed as dist
        own_ms = main_res["own_elapsed"] / args.steps * 1e3  # this rank's own clock around its own steps (before the closing barrier)
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if oversub else f"cuda:{dev_index}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # a few steps more, outside the timed region, with the compute stream stamped around every wait for a panel's exchange
        waits, wl = [], main_res["workload"]
        if isinstance(wl, GemmWorkload) and wl.dist is not None:
            cm = wl.dist["comm"]
            barrier()
            ok = True
            try:
                cm.set_wait_timing(True)
            except Exception as e:  # noqa: BLE001 -- a diagnostic must not take the line down
                ok = False
                log(f"[bench] rank {rank}: wait timing unavailable: {e}")
            # the stamped steps are collective: they run on every rank or on none (a rank stepping alone would leave its peers inside the step's exchange until the
            # launch time-out, with the measured line lost), so the ranks agree first -- and once more afterwards on whether the figures mean anything
            if agree(ok):
                try:
                    for _ in range(3):
                        wl.step()
                    wl.finish()
                    waits = cm.wait_times()
                except Exception as e:  # noqa: BLE001
                    ok = False
                    log(f"[bench] rank {rank}: wait timing failed: {e}")
                if not agree(ok):
                    waits = []
            try:
                cm.set_wait_timing(False)
            except Exception:  # noqa: BLE001
                pass
            barrier()  # (outside the try: every rank passes the same barriers whatever happened to its own diagnostic)
        ck = main_res["clock"]
        per_rank = DIST["all_gather_object"]({"ms_per_step": own_ms, "clock_ghz": (ck or {}).get("mean") if isinstance(ck, dict) else ck, "waits": waits})
        if rank == 0:
            rank_detail, rank_flat = ranks_detail(per_rank, getattr(wl, "npanels", 1))
    w = main_res["workload"]
    global MFMA_CEILING
    if world == 1 and not dist_mode and w.dtype == "f16" and w.bound == "mfma" and os.environ.get("WG_BENCH_NO_CEILING") != "1":
        try:  # right behind the headline's steps, on the same warm chip
            MFMA_CEILING = gpu.mfma_ceiling(0.6)
        except Exception as e:  # noqa: BLE001 -- a diagnostic must not take the line down
            log(f"[bench] wg_debug_mfma_ceiling failed: {e}")
    value, roof = summarize(w, elapsed, main_res["kernel_ms"], args.steps, world, main_res["clock"])
    main_ulp = main_res["max_ulp_vs_f64"]

    main_cpu = main_res["cpu"]
    main_res = None  # release the headline workload's buffers before the secondary configs allocate theirs
    w_name, w_metric, w_unit, w_dtype, w_is_gemm = w.name, w.metric, w.unit, w.dtype, isinstance(w, GemmWorkload)
    par = f"replicas x{world}"
    cfg_extra = {}
    if dist_mode and w_is_gemm:
        engine = {"rccl": "RCCL all-gather", "rccl_cus4": "RCCL all-gather",
                  "staged": "staged peer-copy gather (one SDMA engine per link + relayout)"}[DIST["mode"]]
        par = f"m-shard x{world} + {engine}"
        cfg_extra = {"ranks": dist.get_world_size(), "gather_engine": DIST["mode"], "panel_cols": w.panel_cols, "panels": w.npanels, "panel_tail": ",".join(str(x) for x in (w.panel_widths or [])[-8:]),
                     "all_gather_bytes_per_step": int(w.gather_bytes_per_step()), "stream_compute_units": info.get("stream_compute_units", info["compute_units"]),
                     "comm_compute_units": int(info["compute_units"]) - int(info.get("stream_compute_units", info["compute_units"])),
                     "rccl_reported_ranks": DIST["comm"].reported_size if DIST["comm"].has_collectives else 0,  # ncclCommCount: what RCCL itself says
                     "pipelined_steps": bool(os.environ.get("WG_BENCH_PIPELINED", "1") != "0")}
        if dist_report:
            cfg_extra.update(dist_report)
        cfg_extra.update(rank_flat)
        if oversub:
            cfg_extra["oversubscribed"] = f"{world} ranks on {ndev} GPU(s): plumbing test, not a scaling number"
    if dist_mode and hasattr(w, "close"):
        barrier()
        w.close()  # unmap the peers' buffers before anyone frees them
        barrier()
    w = None
    others = []
    if world == 1 and not dist_mode and not args.no_secondary:
        for name in SECONDARY:
            if name == args.workload or name in args.skip.split(","):
                continue
            try:
                r = run_workload(wg, gpu, name, 10, 3, rank, world, barrier, not args.no_cpu_baseline, min(args.cpu_budget, 5.0),
                                 min_seconds=args.secondary_seconds)
                v, rf = summarize(r["workload"], r["elapsed"], r["kernel_ms"], r["steps"], world, r["clock"])
                others.append({"workload": name, "metric": r["workload"].metric, "value": round(v, 3), "unit": r["workload"].unit,
                               "dtype": r["workload"].dtype, "steps": r["steps"], "roofline": rf, "cpu_baseline": r["cpu"],
                               "max_ulp_vs_f64": None if r["max_ulp_vs_f64"] is None else round(r["max_ulp_vs_f64"], 2)})
            except Exception as e:  # a secondary config must never take the headline down with it
                others.append({"workload": name, "error": f"{type(e).__name__}: {e}"})

    if dist_mode and not args.no_secondary and not oversub:
        import torch
        import torch.distributed as dist
        saved, DIST = DIST, None  # the HBM-bound operators shard by independent units: no communicator
        gpu_all = wg.GpuInstance.new(dev_index)  # its own stream, all CUs: no collective runs next to these

        def barrier_all():
            gpu_all.sync()
            dist.barrier()
            torch.cuda.synchronize()

        for name, nsteps in DIST_SECONDARY:
            if name in args.skip.split(","):
                continue
            err = None
            try:
                r = run_workload(wg, gpu_all, name, nsteps, 20, rank, world, barrier_all, False, 0.0)
                el = r["elapsed"]
            except Exception as e:
                err, el = f"{type(e).__name__}: {e}", float("inf")
            t = torch.tensor([el], dtype=torch.float64, device=f"cuda:{dev_index}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)  # every rank takes part, also one whose run failed
            if not np.isfinite(float(t.item())):
                others.append({"workload": name, "error": err or "failed on another rank"})
                continue
            v, rf = summarize(r["workload"], float(t.item()), r["kernel_ms"], r["steps"], world, r["clock"])
            others.append({"workload": name, "metric": r["workload"].metric, "value": round(v, 3), "unit": r["workload"].unit,
                           "dtype": r["workload"].dtype, "steps": r["steps"], "n_gpus": world, "scaling": "weak",
                           "parallelism": f"row-sharded x{world}, no collective", "roofline": rf})
            r = None
        DIST = saved

    if rank == 0:
        # Every BASELINE config as SCALARS: flat keys in `config` (the driver's parsed record keeps scalar config keys) and, once more, as the
        # compact `targets` object that is the LAST key of the line (the driver's `tail` keeps the last 2000 characters of stdout).
        by = {o["workload"]: o for o in others if "roofline" in o}
        targets = {}

        def put(prefix, o, unit_key, brief=False):  # brief: value + clock only (the tail the driver keeps is 2000 characters)
            if o is None:
                return
            rf = o["roofline"]
            targets[f"{prefix}_{unit_key}"] = round(o["value"], 1)
            if not brief:
                targets[f"{prefix}_frac"] = rf["frac"]
            if "clock_ghz_measured" in rf:
                targets[f"{prefix}_ghz"] = rf["clock_ghz_measured"]
            if "frac_of_ceiling" in rf and not brief:
                targets[f"{prefix}_frac_of_ceiling"] = rf["frac_of_ceiling"]

        head = {"workload": w_name, "value": value, "roofline": roof}
        put("c5_gemm_f16_32768", head if w_name == "gemm_f16_32768" else by.get("gemm_f16_32768"), "tflops")
        put("c5_gemmtr_f16_32768", head if w_name == "gemmtr_f16_32768" else by.get("gemmtr_f16_32768"), "tflops")  # (K-contiguous A: what a rank's shard can be handed as)
        put("c3_gemm_f16_8192", head if w_name == "gemm_f16_8192" else by.get("gemm_f16_8192"), "tflops")
        put("c3_gemmtr_f16_8192", by.get("gemmtr_f16_8192"), "tflops")
        # the reference's own distribution, U[0, 1) (gemm.rs:152), beside the U[-1, 1) headline figures
        put("c5_gemm_f16_32768_u01", by.get("gemm_f16_32768_u01"), "tflops", brief=True)
        put("c3_gemm_f16_8192_u01", by.get("gemm_f16_8192_u01"), "tflops", brief=True)
        put("c3_gemmtr_f16_8192_u01", by.get("gemmtr_f16_8192_u01"), "tflops", brief=True)
        put("c2_gemm_f32_4096_u01", by.get("gemm_f32_4096_u01"), "tflops", brief=True)
        put("c2_gemm_f32_4096", head if w_name == "gemm_f32_4096" else by.get("gemm_f32_4096"), "tflops")
        put("c4_gemv", by.get("gemv_f32_4096x65536"), "gbs")
        put("c4_gemvtr", by.get("gemvtr_f32_65536x4096"), "gbs")
        put("c4_reduce", by.get("reduce_f32_4096x65536"), "gbs")
        put("op_assign", by.get("op_assign_f32_256M"), "gbs", brief=True)
        put("gemm_f16_2048", by.get("gemm_f16_2048"), "tflops", brief=True)
        put("gemm_f16_8192sq_k1024", by.get("gemm_f16_8192x8192x1024"), "tflops", brief=True)
        put("gemmtr_f16_8192sq_k1024", by.get("gemmtr_f16_8192x8192x1024"), "tflops", brief=True)
        put("gemm_f32_2048", by.get("gemm_f32_2048"), "tflops", brief=True)
        put("gemmtr_rm_f16_8192", by.get("gemmtr_rm_f16_8192"), "tflops", brief=True)  # row-major GemmTr (wg_gemm_rm): the NT kernel
        for key, name in (("c1_gemv_1024_us", "gemv_f32_1024"), ("c1_gemv_1024_graph_us", "gemv_f32_1024_graph")):
            if name in by:
                targets[key] = by[name]["roofline"].get("dispatch_us")
        if MFMA_CEILING:
            targets["mfma_only_ceiling_tflops"] = round(MFMA_CEILING["tflops"], 1)
            targets["mfma_only_ceiling_ghz"] = round(MFMA_CEILING["clock_ghz"], 3)
        ulps = {"f16_gemm": [main_ulp] if (w_is_gemm and w_dtype == "f16") else [], "f32_gemm": [main_ulp] if (w_is_gemm and w_dtype == "f32") else [], "f32_gemv": []}
        for o in others:
            u = o.get("max_ulp_vs_f64")
            if u is None:
                continue
            if o["metric"] == "gemm_tflops":
                # U[0,1) operands: every partial sum is positive and grows, rounding errors do not cancel the way they do for U[-1,1): its own figure
                key = ("f16_gemm" if o["dtype"] == "f16" else "f32_gemm") + ("_u01" if o["workload"].endswith("_u01") else "")
                ulps.setdefault(key, []).append(u)
            elif o["metric"] == "gemv_gbs" and o["dtype"] == "f32":
                ulps["f32_gemv"].append(u)
        # sampled entries of every result checked after timing, |gpu - f64| in ulps of the result's format (bounds asserted in tests/test_gpu_ulp.py)
        checks = {f"parity_max_ulp_vs_f64_{k}": round(max(v), 2) for k, v in ulps.items() if v and None not in v}
        targets.update(checks)
        checks["parity_ulp_unit"] = "f32: ulps of sum|a||b| (U[-1,1) operands cancel); f16: ulps of the result (one RNE rounding of an f32 accumulation)"
        cfg_extra.update({k: v for k, v in targets.items() if k.startswith(("c1_", "c2_", "c3_", "c4_", "c5_"))})
        line = {
            "metric": w_metric, "value": round(value, 3), "unit": w_unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "strong" if w_is_gemm else "weak", "vs_baseline": None, "dtype": w_dtype,
            "data": f"synthetic (seeded {'U[0,1)' if VALUES == 'u01' else 'zeros' if VALUES == 'zero' else 'U[-1,1)'}: one 16 Mi-element random block tiled over each operand, resident in HBM "
                    "before the timed region; *_u01 workloads: U[0,1), the reference's new_random)",
            "config": dict({"workload": w_name, "device": info["name"], "compute_units": info["compute_units"], "parallelism": par}, **cfg_extra, **box_facts()),
            "roofline": roof, "cpu_baseline": main_cpu, "checks": checks, "ranks_detail": rank_detail, "others": others,
            "targets": targets,  # LAST key, scalars only: BASELINE configs 1-5 at a glance (value, fraction of the 8 TB/s / 2.5 PF / 157.3 TF peak, measured clock)
        }
        sys.stdout.flush()

        def to_stdout(text):
            if saved_stdout is not None:
                os.dup2(saved_stdout, 1)
            print(text, flush=True)
            if saved_stdout is not None:
                os.dup2(2, 1)  # whatever the libraries print while tearing down goes to stderr as well

        emit(line, args.detail, to_stdout)
    if dist_mode:
        import torch.distributed as dist
        DIST = None
        for g, cm in engines.values():
            cm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
