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
# The product path of the multi-rank run, two processes on ONE GPU: wg_comm (no collective library: RCCL refuses two ranks on one
# device), the output buffers exchanged as IPC handles (wg_buf_ipc_export / _open), wg_gemm_sharded(WG_GATHER_PEER_COPY): every rank's
# Gemm writes its rows of its own M x N C and the copy engine pushes them into the peer's C. Both ranks must hold the plain
# column-major product -- a GpuMatrix any Gemm::dispatch operand can consume (gemm.rs:65-74), which is then fed to one.
# --------------------------------------------------------------------------------------------------------
def _peer_worker(rank, world, port, engine, q):
    try:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        os.environ["WG_PEER_COPY_ENGINE"] = engine
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
        rng = np.random.default_rng(99)
        A = (rng.random((M, K), dtype=np.float32) * 2 - 1).astype(np.float16)
        B = (rng.random((K, N), dtype=np.float32) * 2 - 1).astype(np.float16)
        ta = wg.TensorBuilder.matrix(Mg, K, S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, np.ascontiguousarray(A[rank * Mg:(rank + 1) * Mg].reshape(-1, order="F")))
        tb = wg.TensorBuilder.matrix(K, N, S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, np.ascontiguousarray(B.reshape(-1, order="F")))
        tc = wg.TensorBuilder.matrix(M, N, S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, np.full(M * N, np.nan, np.float16))
        comm = Comm(gpu, world, rank, None)
        handles = [None] * world
        dist.all_gather_object(handles, comm.export_handle(tc))
        comm.register_peers(tc, handles)
        dist.barrier()  # every rank's C is allocated, initialised and mapped before anyone pushes into it
        for _ in range(2):
            comm.sharded_gemm(tc, ta, tb, wg.GemmVariant.Gemm, GatherMode.PEER_COPY, 512)
            gpu.sync()
            comm.flush()
            dist.barrier()
        got = tc.read(dev).reshape(M, N, order="F").astype(np.float64)
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
        dist.barrier()
        q.put((rank, ok and ok2, f"engine {comm.copy_engine}, sent {comm.bytes_sent} B, product ok={ok}, consumable ok={ok2}"))
        comm.close()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc() + str(e)))


@pytest.mark.parametrize("engine", ["sdma", "hip2d"])
def test_two_ranks_one_gpu_peer_copy_gives_a_plain_matrix(engine):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port, world = ctx.Queue(), _free_port(), 2
    procs = [ctx.Process(target=_peer_worker, args=(r, world, port, engine, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, ok, msg in res:
        assert ok, f"rank {rank}: {msg}"
    assert any(("sdma-rect" if engine == "sdma" else "hip2d") in m for _, _, m in res)


# --------------------------------------------------------------------------------------------------------
# The engine for more than two ranks (WG_GATHER_PEER_STAGED), two processes on one GPU: staging cubes + flag arrays exchanged as IPC
# handles, contiguous per-peer copies + sequence-number flags, wait kernel + relayout on the receiving stream -- and NO host
# synchronisation or barrier between three back-to-back steps (stream-ordered, double-buffered by step parity).
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
        comm.flush()
        dist.barrier()
        q.put((rank, ok, f"sent {comm.bytes_sent} B, product ok={ok}"))
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
        for one in (False, True):
            comm.set_one_launch(one)
            tcs = [wg.TensorBuilder.matrix(M, N, U).build_init(dev, np.full(M * N, np.nan, np.float16)) for _ in range(nsteps)]
            for tb, tc in zip(tbs, tcs):
                comm.sharded_gemm(tc, ta, tb, wg.GemmVariant.Gemm, GatherMode.PEER_STAGED, panel)
            comm.join()
            results[one] = [tc.read(dev).view(np.uint16).copy() for tc in tcs]
            comm.flush()
            dist.barrier()
        same = all(np.array_equal(x, y) for x, y in zip(results[False], results[True]))
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
