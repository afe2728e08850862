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


# ------------------------------------------------------------------------------------------------------------
# synthetic data: seeded, U[-1,1) (full-range random operands: zero/constant fills flatter the clocks)
# ------------------------------------------------------------------------------------------------------------
def rand_block(seed: int, n: int, dtype) -> np.ndarray:
    rng = np.random.default_rng(seed)
    if os.environ.get("WG_BENCH_VALUES") == "u01":  # the reference tests' distribution (`new_random`: U[0,1)); experiments only
        return rng.random(n, dtype=np.float32).astype(dtype)
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


class GemmWorkload(Workload):
    bound = "mfma"
    metric = "gemm_tflops"
    unit = "TFLOP/s"

    def __init__(self, name, M, N, K, dtype, trans=False):
        self.name, self.M, self.N, self.K = name, M, N, K
        self.dtype = dtype
        self.trans = trans  # GemmTr: m1 is stored K x M (op(A) = m1^T)
        self.np_dtype = np.float32 if dtype == "f32" else np.float16
        self.kernel = "gemm_f32_kernel" if dtype == "f32" else "gemm_f16_m16_kernel"
        # mid-size outputs: f16 runs the 128 x 128 kernel; f32 cuts K in two and `kernel_ms` spans both launches
        if dtype == "f16" and (M // 256) * (N // 256) < 256 and M * N <= 2560 * 2560:
            self.kernel = "gemm_f16_t128_kernel"
        if dtype == "f32" and (M // 256) * (N // 128) < 256 and M == N == K == 2048:
            self.kernel = "gemm_f32_kernel + splitk_reduce_kernel"

    def setup(self, wg, gpu, rank, world):
        from wgmath_amd.sharded import MShardPlan, ShardedGemm
        self.wg, self.gpu, self.rank, self.world = wg, gpu, rank, world
        # N-panels: the all-gather of panel i overlaps the GEMM of panel i+1 (only meaningful with > 1 rank)
        self.dist_mode = world > 1 or os.environ.get("WG_BENCH_FORCE_DIST") == "1"
        # Panel count: enough panels to overlap the all-gather with compute, but >= ~512 output tiles (256 x 256) per launch:
        # the f16 GEMM workgroup needs a whole CU, so CUs occupied by RCCL's copy kernels are unavailable to it and a panel of
        # exactly 256 tiles (8 ranks: 4096 x 4096 per panel) would take two rounds instead of one.
        tiles = (self.M // max(world, 1) // 256) * (self.N // 256)
        npanels = 1 if not self.dist_mode else max(1, min(8, tiles // 512, self.N // 2048))
        panel_cols = None
        cus = int(gpu.adapter().get("stream_compute_units") or gpu.adapter()["compute_units"])
        if self.dist_mode and self.np_dtype == np.float16 and self.M % (256 * world) == 0 and self.N % 256 == 0:
            # The compute stream may use `cus` CUs (bench main(): the rest is left to RCCL's copy kernels). Size the panels to a whole
            # number of rounds of `cus` workgroups (a 256 x 256 tile each): c tile-columns per panel with the least idle CUs in the
            # last round, between ~1 and ~4 rounds per panel (enough panels to overlap, few enough launches), + one remainder panel.
            tiles_m, tcols = self.M // world // 256, self.N // 256
            best = None
            for c in range(1, tcols + 1):
                t = tiles_m * c
                rounds = -(-t // cus)
                if rounds > 8 or tcols // c < 2:
                    continue
                score = (round(rounds * cus / t, 6), abs(rounds - 2))  # least idle CUs first, then closest to 2 rounds per panel
                if best is None or score < best[0]:
                    best = (score, c)
            if best is not None:
                c = best[1]
                cols = [256 * c] * (tcols // c) + ([256 * (tcols % c)] if tcols % c else [])
                panel_cols, npanels = tuple(cols), len(cols)
        self.plan = plan = MShardPlan(self.M, self.N, self.K, world, npanels, panel_cols)
        self.Mg = plan.Mg  # this rank's rows: A_g = A[g*Mg:(g+1)*Mg, :], its own contiguous col-major tensor
        self.A = device_random(wg, gpu, (self.K, self.Mg) if self.trans else (self.Mg, self.K), self.np_dtype, 0xA000 + rank)
        self.B = device_random(wg, gpu, (self.K, self.N), self.np_dtype, 0xB000)  # replicated
        self.gemm = wg.Gemm.from_device(gpu.device())
        self.shapes = wg.ViewShapeBuffers()
        self.torch_out = None
        S = wg.BufferUsages
        if not self.dist_mode:
            self.C = wg.TensorBuilder.vector(plan.gathered_elems(), S.STORAGE | S.COPY_SRC).build(gpu.device(), self.np_dtype)
        else:
            import torch
            tdt = torch.float32 if self.dtype == "f32" else torch.float16
            # the gathered buffer [panel][rank][np*Mg]; per panel a GpuCube [Mg, np, world] (wgmath_amd/sharded.py)
            self.torch_out = torch.empty(plan.gathered_elems(), dtype=tdt, device=f"cuda:{gpu._ctx.device_index}")
            self.C = wg.GpuTensor.wrap(gpu.device(), self.torch_out.data_ptr(), (plan.gathered_elems(),), self.np_dtype,
                                       keepalive=self.torch_out)
        self.enc = gpu.device().create_command_encoder()
        self.pass_ = self.enc.compute_pass("bench", None)
        a_view = self.A.as_embedded_view(3)

        variant = wg.GemmVariant.GemmTr if self.trans else wg.GemmVariant.Gemm

        def local_gemm(out_shape, a_shape, b_shape):
            self.gemm.dispatch_generic(gpu.device(), self.shapes, self.pass_, wg.GpuTensorView(out_shape, self.C, 2), a_view,
                                       wg.GpuTensorView(b_shape, self.B, 2), variant)

        def all_gather(start, count, rk):
            import torch
            import torch.distributed as dist
            ext = getattr(gpu, "_torch_ext_stream", None)
            if ext is not None:  # the GEMMs run on the library's own (CU-masked) stream: the collective must wait for it
                torch.cuda.current_stream().wait_stream(ext)
            out = self.torch_out[start:start + world * count]
            return dist.all_gather_into_tensor(out, out[rk * count:(rk + 1) * count], async_op=True)

        self.driver = ShardedGemm(plan, rank, local_gemm, all_gather, wait=lambda w: w.wait(), always_gather=self.dist_mode)

    def step(self):
        self.driver.step()
        ext = getattr(self.gpu, "_torch_ext_stream", None)
        if ext is not None:
            # a step is self-contained: the next step's GEMMs (library stream) start after this step's all-gathers (torch's stream,
            # which step() just made wait for them) -- no overlap across steps
            import torch
            ext.wait_stream(torch.cuda.current_stream())

    def units_per_step(self):
        return 2.0 * self.M * self.N * self.K

    def algorithmic_per_launch(self):
        return 2.0 * self.Mg * self.N * self.K / self.plan.npanels  # mean over the launches of a step (one launch = one N-panel of this rank's row block)

    def launches_per_step(self):
        return self.plan.npanels

    def check(self):
        # sampled (row, column-block) entries of this rank's C_g against f64 on the host
        gpu, pl = self.gpu, self.plan
        rng = np.random.default_rng(1)
        rows = np.unique(rng.integers(0, self.Mg, 6))
        cols = np.unique(rng.integers(0, self.N, 24))
        item = np.dtype(self.np_dtype).itemsize
        from wgmath_amd._lib import check, lib
        A = self.A.read(gpu.device())
        A = (A.reshape(self.K, self.Mg, order="F").T if self.trans else A.reshape(self.Mg, self.K, order="F"))[rows].astype(np.float64)

        def read_range(t, start, n):
            out = np.empty(n, self.np_dtype)
            check(lib.wg_buf_read(gpu._ctx.handle, t._h, start * item, out.ctypes.data, n * item))
            return out

        Bc = np.stack([read_range(self.B, c * self.K, self.K) for c in cols], axis=1).astype(np.float64)  # K x ncols
        got = np.empty((rows.size, cols.size))
        for jc, c in enumerate(cols):
            colbuf = read_range(self.C, pl.element_index(self.rank * self.Mg, int(c)), self.Mg)
            got[:, jc] = colbuf[rows]
        truth, sabs = A @ Bc, np.abs(A) @ np.abs(Bc)
        tol = 2 * np.sqrt(self.K) * 2.0 ** -24 * sabs + (2.0 ** -11 * np.abs(truth) + 2.0 ** -25 if self.dtype == "f16" else 0)
        err = np.abs(got - truth)
        assert (err <= tol).all(), f"bench sanity check failed: worst err/tol {(err / tol).max():.3g}"

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
        t0 = time.perf_counter()
        C.gemm(wo.GEMM, out, so, a, s1, b, s2, 0, 1)
        t1 = time.perf_counter() - t0
        n = int(max(1, min(active_wgs, budget_s / max(t1, 1e-3))))
        t0 = time.perf_counter()
        C.gemm(wo.GEMM, out, so, a, s1, b, s2, 0, n)
        dt = time.perf_counter() - t0
        return {"value": flops_per_wg * n / dt / 1e12, "unit": "TFLOP/s", "cores": C.num_threads(), "kind": "port",
                "sample": f"oracle/wgsl_oracle.c `gemm` (f32, naive WGSL order; the reference has no f16 kernel), {min(256, Ms) * n} rows x {Ns} "
                          f"columns x K={K} of the {self.M}x{self.N}x{K} problem, {n} of {active_wgs} active workgroups, {dt:.1f} s"}


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

    def __init__(self, name, R, C, trans, graph_batch=0, nrhs=1):
        self.name, self.R, self.C, self.trans = name, R, C, trans
        self.nrhs = nrhs  # right-hand-side columns (`out_ncols` of the reference: grid.y of gemv.wgsl); the matrix is read ONCE for up to 8
        # launch-bound sizes take the one-kernel path (gemv.hip: rows * cols <= 4 Mi, wgk_gemv)
        self.kernel = "gemv_t_kernel" if trans else ("gemv_n_small_kernel" if R * C <= (4 << 20) and R >= 128 else "gemv_n_kernel")
        # graph_batch > 0: the dispatch is launch-bound (a few MB): record `graph_batch` dispatches into ONE command buffer
        # (a hipGraph) and replay it -- a step is then one Queue::submit of that buffer
        self.graph_batch = graph_batch

    def setup(self, wg, gpu, rank, world):
        self.wg, self.gpu, self.rank, self.world = wg, gpu, rank, world
        R, C = self.R, self.C
        self.m = device_random(wg, gpu, (R, C), np.float32, 0xC000 + rank)
        vlen, olen = (R, C) if self.trans else (C, R)
        S = wg.BufferUsages
        if self.nrhs == 1:
            self.v = device_random(wg, gpu, (vlen,), np.float32, 0xD000)
            self.out = wg.TensorBuilder.vector(olen, S.STORAGE | S.COPY_SRC).build(gpu.device(), np.float32)
        else:
            self.v = device_random(wg, gpu, (vlen, self.nrhs), np.float32, 0xD000)
            self.out = wg.TensorBuilder.matrix(olen, self.nrhs, S.STORAGE | S.COPY_SRC).build(gpu.device(), np.float32)
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
        return 4.0 * (self.R * self.C + self.nrhs * (self.R + self.C))  # matrix + vector(s) + result(s)

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
        tol = 2 * np.sqrt(vlen) * 2.0 ** -24 * sabs
        assert (np.abs(got[idx] - truth) <= tol).all(), "bench sanity check failed (gemv)"

    def cpu_baseline(self, budget_s):
        from oracle import wgsl_oracle as wo
        C = wo.CLib()
        R, Cc = self.R, self.C
        scale = 8  # 1/8 of the columns (or rows): same access pattern, bounded memory (128 MiB) and time
        if self.trans:
            Rs, Cs = R, Cc // scale
        else:
            Rs, Cs = R, Cc // scale
        m = rand_block(3, Rs * Cs, np.float32)
        vlen, olen = (Rs, Cs) if self.trans else (Cs, Rs)
        v, out = rand_block(4, vlen, np.float32), np.zeros(olen, np.float32)
        variant = wo.GEMV_TR if self.trans else wo.GEMV
        C.gemv(variant, out, wo.Shape(olen), m, wo.Shape(Rs, Cs), v, wo.Shape(vlen))  # warm
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < min(budget_s, 10.0) or reps < 3:
            C.gemv(variant, out, wo.Shape(olen), m, wo.Shape(Rs, Cs), v, wo.Shape(vlen))
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        return {"value": 4.0 * (Rs * Cs + Rs + Cs) / dt / 1e9, "unit": "GB/s", "cores": C.num_threads(), "kind": "port",
                "sample": f"oracle/wgsl_oracle.c `{'gemv_tr' if self.trans else 'gemv'}` on a {Rs}x{Cs} slice (1/{scale} of the matrix), "
                          f"mean of {reps} runs, {dt * 1e3:.1f} ms each"}


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
        from oracle import wgsl_oracle as wo
        gpu = self.gpu
        x = self.x.read(gpu.device())
        got = self.res.read(gpu.device())
        for c in (0, 1, self.nvec // 2, self.nvec - 1):
            exp = wo.reduce(wo.SUM, x, wo.Shape(self.n, 1, 1, 1, 1, c * self.n))
            assert np.float32(exp).tobytes() == got[c:c + 1].tobytes(), "bench sanity check failed (reduce is not bit-exact)"

    def cpu_baseline(self, budget_s):
        from oracle import wgsl_oracle as wo
        C = wo.CLib()
        nv = self.nvec // 8
        x = rand_block(5, self.n * nv, np.float32)
        C.reduce_batched(wo.SUM, x, wo.Shape(self.n, nv))
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < min(budget_s, 10.0) or reps < 3:
            C.reduce_batched(wo.SUM, x, wo.Shape(self.n, nv))
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        return {"value": 4.0 * (self.n * nv + nv) / dt / 1e9, "unit": "GB/s", "cores": C.num_threads(), "kind": "port",
                "sample": f"oracle/wgsl_oracle.c reduce (128-lane order) on {nv} of the {self.nvec} vectors, one OpenMP task per vector, "
                          f"mean of {reps} runs, {dt * 1e3:.1f} ms each"}


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
        C.op_assign(wo.ADD, a, wo.Shape(n), b, wo.Shape(n))
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < min(budget_s, 5.0) or reps < 3:
            C.op_assign(wo.ADD, a, wo.Shape(n), b, wo.Shape(n))
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        return {"value": 12.0 * n / dt / 1e9, "unit": "GB/s", "cores": C.num_threads(), "kind": "port",
                "sample": f"oracle/wgsl_oracle.c op_assign(Add) on {n} elements, mean of {reps} runs, {dt * 1e3:.1f} ms each"}


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
    "gemmtr_f32_4096": lambda: GemmWorkload("gemmtr_f32_4096", 4096, 4096, 4096, "f32", trans=True),
    "gemv_f32_4096x65536": lambda: GemvWorkload("gemv_f32_4096x65536", 4096, 65536, False),
    "gemvtr_f32_65536x4096": lambda: GemvWorkload("gemvtr_f32_65536x4096", 65536, 4096, True),
    "gemv_f32_4096x65536_rhs8": lambda: GemvWorkload("gemv_f32_4096x65536_rhs8", 4096, 65536, False, nrhs=8),
    "gemv_f32_1024": lambda: GemvWorkload("gemv_f32_1024", 1024, 1024, False),
    "gemv_f32_1024_graph": lambda: GemvWorkload("gemv_f32_1024_graph", 1024, 1024, False, graph_batch=64),
    "reduce_f32_4096x65536": lambda: ReduceWorkload("reduce_f32_4096x65536", 4096, 65536),
    "op_assign_f32_256M": lambda: OpAssignWorkload("op_assign_f32_256M", 1 << 28),
}
# Headline: the north-star's M-sharded f16 GEMM (BASELINE configs[4]); the SAME problem at every --gpus N ("strong"), so the
# driver's per-N values are comparable. It fits one GPU (3 x 2 GiB), which makes it the N = 1 workload as well.
DEFAULT_WORKLOAD = "gemm_f16_32768"
SECONDARY = ["gemm_f16_8192", "gemm_f32_4096", "gemm_f16_2048", "gemm_f32_2048", "gemm_f16_ts_131072x1024x8192", "gemm_f32_ts_65536x512x4096", "gemm_f32_fewcols_32000x16x4096", "gemv_f32_4096x65536", "gemvtr_f32_65536x4096", "gemv_f32_4096x65536_rhs8", "reduce_f32_4096x65536",
             "op_assign_f32_256M", "gemv_f32_1024", "gemv_f32_1024_graph"]


# With N > 1 ranks the HBM-bound operators shard by independent units with no data-path collective: every rank
# streams its own row block of an N-times taller matrix ("weak"); (name, fixed step count -- the same on every rank).
DIST_SECONDARY = [("gemv_f32_4096x65536", 2000), ("gemvtr_f32_65536x4096", 2000), ("reduce_f32_4096x65536", 2000)]


def load_traffic(workload: str):
    """HBM bytes per launch measured with rocprofv3 --pmc (separate passes), recorded under profiles/ (DESIGN.md)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path)).get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def run_workload(wg, gpu, name, steps, warmup, rank, world, barrier, with_cpu, cpu_budget, min_seconds=0.0):
    """Times EXACTLY `steps` steps (after `warmup` untimed ones). With min_seconds > 0 (secondary configs only) `steps` is raised
    so that the timed region lasts at least that long: millisecond kernels timed for a few tens of ms still see the clocks
    ramping (measured: f32 GEMM 140 TF over 30 ms vs 148 TF sustained)."""
    w = WORKLOADS[name]()
    w.setup(wg, gpu, rank, world)
    t_w = time.perf_counter()
    for _ in range(max(warmup, 1)):
        w.step()
    gpu.sync()
    if min_seconds > 0:
        per_step = max((time.perf_counter() - t_w) / max(warmup, 1), 1e-6)
        steps = int(min(max(steps, min_seconds / per_step), 20000))
        for _ in range(min(steps, 50)):  # a little more warm-up at the final cadence
            w.step()
    ts = wg.GpuTimestamps.new(gpu.device(), 2)
    barrier()
    gpu.sync()
    t0 = time.perf_counter()
    ts.write(gpu.device())
    for _ in range(steps):
        w.step()
    ts.write(gpu.device())
    gpu.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    ev = ts.wait_for_results_ms()
    launches = getattr(w, "launches_per_step", lambda: 1)()
    kernel_ms = (ev[1] - ev[0]) / (steps * launches)  # HIP events on the stream the kernels run on
    if not os.environ.get("WG_BENCH_NO_CHECK"):
        w.check()
    res = {"workload": w, "elapsed": elapsed, "kernel_ms": kernel_ms, "steps": steps}
    res["cpu"] = w.cpu_baseline(cpu_budget) if (with_cpu and rank == 0 and world == 1) else None
    return res


def summarize(w, elapsed, kernel_ms, steps, world):
    scale = 1e12 if w.unit == "TFLOP/s" else 1e9
    value = w.units_per_step() * steps / elapsed / scale
    achieved = w.algorithmic_per_launch() / (kernel_ms * 1e-3) / scale
    peak = PEAK_MFMA_TFLOPS[w.dtype] if w.bound == "mfma" else PEAK_HBM_GBS
    # the PMC traffic figure was measured on the single-GPU, one-launch-per-step form of the workload: null for any other launch shape
    launches = getattr(w, "launches_per_step", lambda: 1)()
    traffic = load_traffic(w.name) if (world == 1 and launches in (1, getattr(w, "graph_batch", 0))) else None
    roof = {"bound": w.bound, "kernel": w.kernel, "achieved": round(achieved, 3), "peak": peak, "unit": w.unit,
            "frac": round(achieved / peak, 4), "kernel_ms": round(kernel_ms, 5), "traffic": traffic}
    return value, roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra single-GPU configs reported under `others`")
    ap.add_argument("--skip", default="", help="comma-separated secondary workloads to skip (e.g. under rocprofv3)")
    ap.add_argument("--secondary-seconds", type=float, default=0.6, help="minimum timed duration of each secondary config")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU work for the cpu_baseline sample")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs one process per GPU: launch with "
                     f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py ...`")
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # WG_BENCH_FORCE_DIST=1: take the torch.distributed/RCCL path even with one rank (single-GPU test of the N > 1 plumbing)
    dist_mode = world > 1 or os.environ.get("WG_BENCH_FORCE_DIST") == "1"
    if dist_mode:
        # torch bundles its own ROCm runtime (same soname as /opt/rocm's): it must be loaded FIRST so that libwgebra_hip.so
        # binds to that one copy -- two HIP runtimes in one process cannot both drive the GPU (wgmath_amd checks for this)
        import torch  # noqa: F401
    import wgmath_amd as wg

    if dist_mode:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch
        import torch.distributed as dist
        # CU partitioning between compute and communication. Every f16 GEMM workgroup needs a whole CU (160 KiB LDS, 512 registers per
        # lane), so a copy kernel of the collective library, launched from a second queue while a GEMM grid is resident, only gets CUs
        # when GEMM workgroups retire -- measured with tools/overlap_probe.py: its workgroups start ~0.75-1.4 ms late, i.e. the
        # all-gather of panel i would barely overlap the GEMM of panel i+1. So the GEMM stream is CU-masked to leave `comm_cus` CUs
        # free: the copy kernels (small workgroups, several fit on one CU) then start within ~20 us of their launch. 32 = one CU per
        # shader engine of every XCD (a mask that is not a multiple of 32 unbalances the shader engines: 240 CUs ran 20 % slower than
        # 224). Per tile the GEMM is ~4 % slower on 224 CUs than on 256 shared with a copy kernel (tools/overlap_probe2.py).
        comm_cus = int(os.environ.get("WG_BENCH_COMM_CUS", "32"))
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        total_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
        gpu = None
        if comm_cus > 0 and total_cus > 2 * comm_cus:
            try:
                gpu = wg.GpuInstance.new(local_rank, cu_count=total_cus - comm_cus)
                gpu._torch_ext_stream = torch.cuda.ExternalStream(gpu.stream(), device=torch.device(f"cuda:{local_rank}"))
            except Exception as e:  # no CU-masked stream on this runtime: share torch's stream (the transport is unchanged)
                log(f"[bench] CU-masked compute stream unavailable ({e}); sharing torch's stream")
                gpu = None
        if gpu is None:
            # share torch's current stream so that the GEMM and the all-gather are ordered without extra events
            gpu = wg.GpuInstance.new(local_rank, stream=torch.cuda.current_stream().cuda_stream)

        def barrier():
            gpu.sync()
            dist.barrier()
            torch.cuda.synchronize()
    else:
        gpu = wg.GpuInstance.new(local_rank)

        def barrier():
            gpu.sync()

    info = gpu.adapter()
    _pad = None
    if os.environ.get("WG_BENCH_PAD"):  # experiment hook: shift every later allocation by this many bytes
        _pad = wg.TensorBuilder.vector(int(os.environ["WG_BENCH_PAD"]) // 4, wg.BufferUsages.STORAGE).build(gpu.device(), np.float32)
    main_res = run_workload(wg, gpu, args.workload, args.steps, args.warmup, rank, world, barrier,
                            not args.no_cpu_baseline, args.cpu_budget)
    elapsed = main_res["elapsed"]
    if dist_mode:
        import torch
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    w = main_res["workload"]
    value, roof = summarize(w, elapsed, main_res["kernel_ms"], args.steps, world)

    main_cpu = main_res["cpu"]
    main_res = None  # release the headline workload's buffers before the secondary configs allocate theirs
    w_name, w_metric, w_unit, w_dtype, w_is_gemm = w.name, w.metric, w.unit, w.dtype, isinstance(w, GemmWorkload)
    w = None
    others = []
    if world == 1 and not args.no_secondary:
        for name in SECONDARY:
            if name == args.workload or name in args.skip.split(","):
                continue
            try:
                r = run_workload(wg, gpu, name, 10, 3, rank, world, barrier, not args.no_cpu_baseline, min(args.cpu_budget, 5.0),
                                 min_seconds=args.secondary_seconds)
                v, rf = summarize(r["workload"], r["elapsed"], r["kernel_ms"], r["steps"], world)
                others.append({"workload": name, "metric": r["workload"].metric, "value": round(v, 3), "unit": r["workload"].unit,
                               "dtype": r["workload"].dtype, "steps": r["steps"], "roofline": rf, "cpu_baseline": r["cpu"]})
            except Exception as e:  # a secondary config must never take the headline down with it
                others.append({"workload": name, "error": f"{type(e).__name__}: {e}"})

    if dist_mode and not args.no_secondary:
        import torch
        import torch.distributed as dist
        gpu_all = wg.GpuInstance.new(local_rank)  # its own stream, all CUs: no collective runs next to these

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
            t = torch.tensor([el], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)  # every rank takes part, also one whose run failed
            if not np.isfinite(float(t.item())):
                others.append({"workload": name, "error": err or "failed on another rank"})
                continue
            v, rf = summarize(r["workload"], float(t.item()), r["kernel_ms"], r["steps"], world)
            others.append({"workload": name, "metric": r["workload"].metric, "value": round(v, 3), "unit": r["workload"].unit,
                           "dtype": r["workload"].dtype, "steps": r["steps"], "n_gpus": world, "scaling": "weak",
                           "parallelism": f"row-sharded x{world}, no collective", "roofline": rf})
            r = None

    if rank == 0:
        line = {
            "metric": w_metric, "value": round(value, 3), "unit": w_unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "strong" if w_is_gemm else "weak", "vs_baseline": None, "dtype": w_dtype,
            "data": "synthetic (seeded U[-1,1), resident in HBM before the timed region)",
            "config": {"workload": w_name, "device": info["name"], "compute_units": info["compute_units"],
                       "parallelism": f"m-shard x{world} + RCCL all-gather" if (world > 1 and w_is_gemm) else f"replicas x{world}"},
            "roofline": roof, "cpu_baseline": main_cpu, "others": others,
        }
        print(json.dumps(line), flush=True)
    if dist_mode:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
