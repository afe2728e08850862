"""Two ranks of the M-sharded GEMM on ONE GPU (both processes use device 0): the real HIP kernels, the real MShardPlan /
ShardedGemm driver with ragged panels and a CU-masked compute stream per rank, rank-dependent addressing included; only the
transport is a stand-in (gloo all-gather of host copies -- RCCL refuses two ranks on one device). The multi-GPU bench path
(bench.py --gpus N) differs from this in the collective alone."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    try:
        import torch  # first: one HIP runtime per process
        import torch.distributed as dist
        import wgmath_amd as wg
        from wgmath_amd._lib import check, lib
        from wgmath_amd.sharded import MShardPlan, ShardedGemm
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        M, N, K = 1024, 1536, 512
        pl = MShardPlan(M, N, K, world, panel_cols=(512, 512, 256, 256))
        gpu = wg.GpuInstance.new(0, cu_count=224)          # what bench.py --gpus N does: leave 32 CUs to the copy kernels
        dev, S = gpu.device(), wg.BufferUsages
        rng = np.random.default_rng(99)                    # same A, B everywhere; every rank keeps its row block
        A = (rng.random((M, K), dtype=np.float32) * 2 - 1).astype(np.float16)
        B = (rng.random((K, N), dtype=np.float32) * 2 - 1).astype(np.float16)
        r0, nr = pl.a_rows(rank)
        ta = wg.TensorBuilder.tensor((nr, K), S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, np.ascontiguousarray(A[r0:r0 + nr].reshape(-1, order="F")))
        tb = wg.TensorBuilder.tensor((K, N), S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, np.ascontiguousarray(B.reshape(-1, order="F")))
        tc = wg.TensorBuilder.vector(pl.gathered_elems(), S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, np.full(pl.gathered_elems(), np.nan, np.float16))
        gemm, shapes = wg.Gemm.from_device(dev), wg.ViewShapeBuffers()
        pass_ = dev.create_command_encoder().compute_pass("dist2", None)
        a_view = ta.as_embedded_view(3)

        def local_gemm(out_shape, a_shape, b_shape):
            gemm.dispatch(dev, shapes, pass_, wg.GpuTensorView(out_shape, tc, 2), a_view, wg.GpuTensorView(b_shape, tb, 2))

        def all_gather(start, count, rk):  # host-staged stand-in for the in-place RCCL all-gather of [start, start + world*count)
            gpu.sync()
            mine = np.empty(count, np.float16)
            check(lib.wg_buf_read(gpu._ctx.handle, tc._h, (start + rk * count) * 2, mine.ctypes.data, count * 2))
            outs = [torch.empty(count, dtype=torch.float16) for _ in range(world)]
            dist.all_gather(outs, torch.from_numpy(mine))
            for g, t in enumerate(outs):
                if g != rk:
                    arr = t.numpy()
                    check(lib.wg_buf_write(gpu._ctx.handle, tc._h, (start + g * count) * 2, arr.ctypes.data, count * 2))
            return None

        ShardedGemm(pl, rank, local_gemm, all_gather).step()
        gpu.sync()
        got_flat = tc.read(dev)
        assert not np.isnan(got_flat.astype(np.float32)).any(), "holes in the gathered buffer"
        truth = A.astype(np.float64) @ B.astype(np.float64)
        sabs = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
        got = np.empty((M, N))
        for p in range(pl.npanels):
            cube = pl.cube_shape(p)
            nc, c0 = pl.cols_of(p), pl.col0_of(p)
            for g in range(world):
                idx = cube.offset + g * cube.stride_mat + np.arange(pl.Mg)[:, None] + np.arange(nc)[None, :] * cube.stride
                got[g * pl.Mg:(g + 1) * pl.Mg, c0:c0 + nc] = got_flat[idx]
        tol = 2 * np.sqrt(K) * 2.0 ** -24 * sabs + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25
        ok = bool((np.abs(got - truth) <= tol).all())
        q.put((rank, ok, "" if ok else f"worst err/tol {(np.abs(got - truth) / tol).max():.3g}"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


def test_sharded_gemm_two_ranks_one_gpu():
    import multiprocessing as mp  # not torch.multiprocessing: the pytest process must not load torch's bundled HIP runtime next to the library's
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in results), results


# --------------------------------------------------------------------------------------------------------
# The copy-engine exchange (WG_GATHER_PEER_STAGED), two processes on one GPU (RCCL refuses two ranks on one device): staging cubes +
# flag arrays exchanged as IPC handles (wg_buf_ipc_export / _open), contiguous per-peer copies + sequence-number flags, wait kernel +
# relayout on the receiving stream -- and NO host synchronisation or barrier between three back-to-back steps (stream-ordered,
# double-buffered by step parity). Both ranks must hold the plain column-major product -- a GpuMatrix any Gemm::dispatch operand can
# consume (gemm.rs:65-74), which is then fed to one.
# --------------------------------------------------------------------------------------------------------
def _staged_worker(rank, world, port, q):
    try:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        os.environ["GPU_MAX_HW_QUEUES"] = "16"  # one hardware queue per stream: copies run beside the Gemms, not behind them
        import torch  # first: one HIP runtime per process
        import torch.distributed as dist
        import wgmath_amd as wg
        from wgmath_amd.sharded import Comm, GatherMode
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        M, N, K = 1024, 1536, 512
        Mg = M // world
        gpu = wg.GpuInstance.new(0)
        dev, S = gpu.device(), wg.BufferUsages
        rng = np.random.default_rng(77)
        A = (rng.random((M, K), dtype=np.float32) * 2 - 1).astype(np.float16)
        B = (rng.random((K, N), dtype=np.float32) * 2 - 1).astype(np.float16)
        ta = wg.TensorBuilder.matrix(Mg, K, S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, np.ascontiguousarray(A[rank * Mg:(rank + 1) * Mg].reshape(-1, order="F")))
        tb = wg.TensorBuilder.matrix(K, N, S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, np.ascontiguousarray(B.reshape(-1, order="F")))
        tc = wg.TensorBuilder.matrix(M, N, S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, np.full(M * N, np.nan, np.float16))
        comm = Comm(gpu, world, rank, None)
        pairs = [None] * world
        dist.all_gather_object(pairs, comm.stage_export(2 * M * N * 2))
        comm.set_peer_stages(pairs)
        dist.barrier()  # every rank's cubes and flags exist and are mapped before anyone pushes
        for pipelined in (False, True):
            comm.set_pipelined(pipelined)  # True: each call's last panel completes inside the next call (or join)
            for _ in range(3):
                comm.sharded_gemm(tc, ta, tb, wg.GemmVariant.Gemm, GatherMode.PEER_STAGED, 512)
            comm.join()
        got = tc.read(dev).reshape(M, N, order="F").astype(np.float64)  # stream order is all it takes
        A64, B64 = A.astype(np.float64), B.astype(np.float64)
        truth, sabs = A64 @ B64, np.abs(A64) @ np.abs(B64)
        tol = 2 * 2.0 * np.sqrt(K) * 2.0 ** -24 * sabs + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25
        ok = bool((np.abs(got - truth) <= tol).all())
        # the gathered C is an ordinary GpuMatrix: use it as the m1 of another Gemm (C^T C would overflow f16: scale by a thin m2)
        thin = (rng.random((N, 8), dtype=np.float32) / N).astype(np.float16)
        tt = wg.TensorBuilder.matrix(N, 8, S.STORAGE | S.COPY_DST).build_init(dev, np.ascontiguousarray(thin.reshape(-1, order="F")))
        to = wg.TensorBuilder.matrix(M, 8, S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, np.zeros(M * 8, np.float16))
        p = dev.create_command_encoder().compute_pass("use", None)
        wg.Gemm.from_device(dev).dispatch(dev, wg.ViewShapeBuffers(), p, to, tc, tt)
        used = to.read(dev).reshape(M, 8, order="F").astype(np.float64)
        ref = tc.read(dev).reshape(M, N, order="F").astype(np.float64) @ thin.astype(np.float64)
        ok2 = bool(np.abs(used - ref).max() <= 2.0 ** -9 * np.abs(ref).max() + 1e-3)
        comm.flush()
        dist.barrier()
        q.put((rank, ok and ok2, f"sent {comm.bytes_sent} B, product ok={ok}, consumable ok={ok2}"))
        comm.close()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc() + str(e)))


def test_two_ranks_one_gpu_staged_peer_copies():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port, world = ctx.Queue(), _free_port(), 2
    procs = [ctx.Process(target=_staged_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, ok, msg in res:
        assert ok, f"rank {rank}: {msg}"


# --------------------------------------------------------------------------------------------------------
# Staged engine, operands that CHANGE every step and shapes that change between steps, pipelined, no host synchronisation anywhere:
# a slot overwritten while the peer still reads it (single-panel steps used to run two steps ahead), or a parity half that moved with
# the shape, shows up as wrong rows in one of the per-step outputs. Every step has its own B and its own output tensor.
# --------------------------------------------------------------------------------------------------------
def _staged_steps_worker(rank, world, port, q):
    try:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        os.environ["GPU_MAX_HW_QUEUES"] = "16"
        import torch  # noqa: F401  first: one HIP runtime per process
        import torch.distributed as dist
        import wgmath_amd as wg
        from wgmath_amd.sharded import Comm, GatherMode
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        M, K = 1024, 512
        Mg = M // world
        # (N, panel_cols): one-panel steps back to back (the case that ran ahead), several panels, and a change of shape in between
        plan = [(1536, 1536), (1536, 1536), (1536, 1536), (1536, 1536), (1024, 512), (1024, 512), (1536, 1536), (1536, 512), (512, 512), (512, 512)]
        gpu = wg.GpuInstance.new(0)
        dev, S = gpu.device(), wg.BufferUsages
        U = S.STORAGE | S.COPY_SRC | S.COPY_DST
        rng = np.random.default_rng(123)
        A = (rng.random((M, K), dtype=np.float32) * 2 - 1).astype(np.float16)
        ta = wg.TensorBuilder.matrix(Mg, K, U).build_init(dev, np.ascontiguousarray(A[rank * Mg:(rank + 1) * Mg].reshape(-1, order="F")))
        Bs, tbs, tcs = [], [], []
        for (N, _) in plan:
            B = (rng.random((K, N), dtype=np.float32) * 2 - 1).astype(np.float16)
            Bs.append(B)
            tbs.append(wg.TensorBuilder.matrix(K, N, U).build_init(dev, np.ascontiguousarray(B.reshape(-1, order="F"))))
            tcs.append(wg.TensorBuilder.matrix(M, N, U).build_init(dev, np.full(M * N, np.nan, np.float16)))
        comm = Comm(gpu, world, rank, None)
        pairs = [None] * world
        dist.all_gather_object(pairs, comm.stage_export(2 * M * 1536 * 2))
        comm.set_peer_stages(pairs)
        dist.barrier()
        comm.set_pipelined(True)
        for (N, pc), tb, tc in zip(plan, tbs, tcs):
            comm.sharded_gemm(tc, ta, tb, wg.GemmVariant.Gemm, GatherMode.PEER_STAGED, pc)
        comm.join()
        bad = []
        A64 = A.astype(np.float64)
        for s, ((N, pc), B, tc) in enumerate(zip(plan, Bs, tcs)):
            got = tc.read(dev).reshape(M, N, order="F").astype(np.float64)
            B64 = B.astype(np.float64)
            truth, sabs = A64 @ B64, np.abs(A64) @ np.abs(B64)
            tol = 2 * 2.0 * np.sqrt(K) * 2.0 ** -24 * sabs + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25
            if not bool((np.abs(got - truth) <= tol).all()):
                bad.append(s)
        comm.flush()
        dist.barrier()
        q.put((rank, not bad, f"steps with wrong rows: {bad}"))
        comm.close()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc() + str(e)))


def test_two_ranks_one_gpu_staged_steps_with_changing_operands_and_shapes():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port, world = ctx.Queue(), _free_port(), 2
    procs = [ctx.Process(target=_staged_steps_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, ok, msg in res:
        assert ok, f"rank {rank}: {msg}"


# --------------------------------------------------------------------------------------------------------
# One launch per step with two ranks (two processes, one GPU): each rank's whole product is ONE kernel over all N-panels that raises a flag
# per panel; the peer streams wait on the flags (hipStreamWaitValue32) and push the slots while the kernel is still running; operands
# change every step, pipelined, no host synchronisation. Must equal the panel-by-panel launches bit for bit.
# --------------------------------------------------------------------------------------------------------
def _staged_one_launch_worker(rank, world, port, q):
    try:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        os.environ["GPU_MAX_HW_QUEUES"] = "16"
        import torch  # noqa: F401  first: one HIP runtime per process
        import torch.distributed as dist
        import wgmath_amd as wg
        from wgmath_amd.sharded import Comm, GatherMode
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        M, K, N, panel = 4096, 512, 8192 + 256, 2048  # 2048-row shards: 8 x 33 tiles >= 256; a ragged last panel
        Mg = M // world
        gpu = wg.GpuInstance.new(0)
        dev, S = gpu.device(), wg.BufferUsages
        U = S.STORAGE | S.COPY_SRC | S.COPY_DST
        rng = np.random.default_rng(2024)
        A = (rng.random((M, K), dtype=np.float32) * 2 - 1).astype(np.float16)
        ta = wg.TensorBuilder.matrix(Mg, K, U).build_init(dev, np.ascontiguousarray(A[rank * Mg:(rank + 1) * Mg].reshape(-1, order="F")))
        nsteps = 4
        Bs = [(rng.random((K, N), dtype=np.float32) * 2 - 1).astype(np.float16) for _ in range(nsteps)]
        tbs = [wg.TensorBuilder.matrix(K, N, U).build_init(dev, np.ascontiguousarray(B.reshape(-1, order="F"))) for B in Bs]
        comm = Comm(gpu, world, rank, None)
        pairs = [None] * world
        dist.all_gather_object(pairs, comm.stage_export(2 * M * N * 2))
        comm.set_peer_stages(pairs)
        dist.barrier()
        comm.set_pipelined(True)
        results = {}
        # "taper": the one-launch form on a tapered tail (wg_gemm_sharded_panels): two equal panels, then 6, 4, 3, 2, 1, 1 tile columns -- the same
        # tiles and chains as any other split, so the same bits; the slot layout (and so the peers' copies) is another one
        taper = [2048, 2048, 1536, 1024, 768, 512, 256, 256]
        assert sum(taper) == N
        for one in (False, True, "taper"):
            comm.set_one_launch(bool(one))
            tcs = [wg.TensorBuilder.matrix(M, N, U).build_init(dev, np.full(M * N, np.nan, np.float16)) for _ in range(nsteps)]
            for tb, tc in zip(tbs, tcs):
                comm.sharded_gemm(tc, ta, tb, wg.GemmVariant.Gemm, GatherMode.PEER_STAGED, taper if one == "taper" else panel)
            comm.join()
            results[one] = [tc.read(dev).view(np.uint16).copy() for tc in tcs]
            comm.flush()
            dist.barrier()
        same = all(np.array_equal(x, y) for x, y in zip(results[False], results[True])) and all(np.array_equal(x, y) for x, y in zip(results["taper"], results[True]))
        A64 = A.astype(np.float64)
        rows = np.unique(rng.integers(0, M, 64))
        ok = True
        for B, r in zip(Bs, results[True]):
            got = r.view(np.float16).reshape(N, M).T[rows].astype(np.float64)
            B64 = B.astype(np.float64)
            truth, sabs = A64[rows] @ B64, np.abs(A64[rows]) @ np.abs(B64)
            tol = 2 * 2.0 * np.sqrt(K) * 2.0 ** -24 * sabs + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25
            ok = ok and bool((np.abs(got - truth) <= tol).all())
        q.put((rank, ok and same, f"values ok={ok}, one launch == panel launches: {same}"))
        comm.close()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc() + str(e)))


def test_two_ranks_one_gpu_one_launch_per_step():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port, world = ctx.Queue(), _free_port(), 2
    procs = [ctx.Process(target=_staged_one_launch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, ok, msg in res:
        assert ok, f"rank {rank}: {msg}"


# --------------------------------------------------------------------------------------------------------
# BASELINE config 5 at FULL size with two ranks: f16 32768^3, M-sharded over two processes that share the one GPU, through the bench's own
# launcher (one JSON line, rank 0's sanity check reads rows the OTHER rank computed). The size at which the removed SDMA rect-copy engine
# used to hang; the staged engine must finish well inside the limit.
# --------------------------------------------------------------------------------------------------------
def test_two_processes_one_gpu_config5_full_size_staged():
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WG_BENCH_OVERSUBSCRIBE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", WG_BENCH_LAUNCH_TIMEOUT="600")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--gather", "staged", "--steps", "2", "--warmup", "1",
                        "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["workload"] == "gemm_f16_32768" and line["config"]["gather_engine"] == "staged"
    assert line["config"]["all_gather_bytes_per_step"] == 16384 * 32768 * 2 and line["value"] > 100.0


# --------------------------------------------------------------------------------------------------------
# Communicators of 2, 4 and 8 emulated ranks created, used and destroyed back to back in ONE process (tools/rank_emulation.py: the staged engine's compute side
# panel by panel and as one launch per step, then a 248-CU context with its own communicator, uniform and tapered panels, for every rank count) -- the sequence
# a `bench.py --gpus 1,2,4,8` session walks through, and the one that once did not return for 30 minutes (profiles/r05_evidence.md section 4; not reproduced
# since: profiles/r06_evidence.md). A fresh child under a hard limit; the tool prints where it stands should it ever stop again.
# --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("size", ["8192"])
def test_communicators_of_2_4_8_emulated_ranks_back_to_back_in_one_process(size):
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EMU_SIZE=size, EMU_TIMEOUT="100", STEPS="3")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rank_emulation.py"), "2", "4", "8", "2", "8", "4"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    assert time.time() - t0 < 120
    out = json.loads(r.stdout)
    assert set(out["ranks"]) == {"2", "4", "8"} and out["single_gpu_best"]["tflops"] > 100
    for P, res in out["ranks"].items():
        for key in ("no_exchange", "staged_compute_and_relayout_panel_launches", "staged_compute_and_relayout", "rccl_compute_side_248_cus", "rccl_compute_side_248_cus_tapered"):
            assert res[key]["ms_per_step"] > 0, (P, key)


# --------------------------------------------------------------------------------------------------------
# The bench line itself (tests/test_bench_launcher.py checks the launcher on CPU; these need the GPU): the single-GPU line ends with the scalar
# `targets` object and carries the clock measured in the run + the MFMA-only ceiling; the communicator path (WG_BENCH_FORCE_DIST=1: RCCL with
# one rank) reports the rank count RCCL itself gives and the CU split.
# --------------------------------------------------------------------------------------------------------
def _bench_line(*argv, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_line_ends_with_targets_and_carries_the_measured_clock():
    line = _bench_line("--workload", "gemm_f16_8192", "--steps", "20", "--warmup", "5", "--no-secondary", "--no-cpu-baseline")
    assert list(line)[-1] == "targets" and all(not isinstance(v, (dict, list)) for v in line["targets"].values())
    rf, t = line["roofline"], line["targets"]
    assert 0.8 < rf["clock_ghz_measured"] < 2.6 and rf["clock_ghz_xcd_min_max"][0] <= rf["clock_ghz_measured"] <= rf["clock_ghz_xcd_min_max"][1]
    assert 1200.0 < rf["mfma_only_ceiling_tflops"] < 2600.0 and 0.3 < rf["frac_of_ceiling"] < 1.05
    assert abs(rf["frac_at_measured_clock"] - rf["frac"] * 2.4 / rf["clock_ghz_measured"]) < 2e-3
    assert t["c3_gemm_f16_8192_tflops"] == round(line["value"], 1) and t["c3_gemm_f16_8192_frac"] == rf["frac"] and t["c3_gemm_f16_8192_ghz"] == rf["clock_ghz_measured"]
    assert line["config"]["c3_gemm_f16_8192_tflops"] == t["c3_gemm_f16_8192_tflops"]  # flat scalars in `config` too
    assert 0 < line["checks"]["parity_max_ulp_vs_f64_f16_gemm"] <= 1.0  # one rounding of an f32 accumulation: within an f16 ulp of f64


def test_default_bench_line_fits_the_driver_and_the_sidecar_holds_the_rest(tmp_path):
    """The DEFAULT workload set (headline f16 32768^3 + every secondary config, cpu_baseline on): what the driver runs at round end, with short
    timed regions only. The stdout line must stay <= 8 KB of strict JSON (round 5's 25 KB line came back unparsed) and still carry `roofline`,
    `cpu_baseline` and the C1-C5 scalars; `others` lives in the sidecar."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    side = tmp_path / "detail.json"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--secondary-seconds", "0.05", "--cpu-budget", "2",
                        "--detail", str(side)], capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 8192, (len(lines), len(lines[0]))

    def bad(tok):
        raise ValueError(f"non-standard JSON constant {tok}")
    line = json.loads(lines[0], parse_constant=bad)
    assert "others" not in line and list(line)[-1] == "targets"
    assert line["config"]["workload"] == "gemm_f16_32768" and line["dtype"] == "f16" and line["n_gpus"] == 1 and line["steps"] == 3
    assert line["roofline"]["bound"] == "mfma" and 0.2 < line["roofline"]["frac"] < 1.0 and line["roofline"]["peak"] == 2500.0
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
    t = line["targets"]
    for k in ("c5_gemm_f16_32768_tflops", "c5_gemmtr_f16_32768_tflops", "c3_gemm_f16_8192_tflops", "c3_gemmtr_f16_8192_tflops", "c2_gemm_f32_4096_tflops",
              "c4_gemv_gbs", "c4_gemvtr_gbs", "c4_reduce_gbs", "c1_gemv_1024_us", "c1_gemv_1024_graph_us"):
        assert t[k] > 0 and line["config"][k] == t[k], k
    full = json.loads(side.read_text(), parse_constant=bad)
    errs = [o for o in full["others"] if "error" in o]
    assert not errs, errs
    assert len(full["others"]) >= 24 and all("roofline" in o for o in full["others"]) and full["targets"] == t


def test_bench_communicator_path_reports_rccl_rank_count_and_cu_split():
    line = _bench_line("--workload", "gemm_f16_8192", "--gather", "rccl", "--steps", "3", "--warmup", "1", "--no-secondary", "--no-cpu-baseline",
                       env={"WG_BENCH_FORCE_DIST": "1"})
    cfg = line["config"]
    assert cfg["gather_engine"] == "rccl" and cfg["ranks"] == 1
    assert cfg["rccl_reported_ranks"] == 1  # ncclCommCount through wg_comm_reported_size
    assert cfg["comm_compute_units"] == 0 and cfg["stream_compute_units"] == cfg["compute_units"]  # one rank: nothing to leave to RCCL
    assert list(line)[-1] == "targets"
