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
    import torch.multiprocessing as mp
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
