"""The N > 1 path on CPU: world-size-2 `gloo` processes run the M-shard planner + pipelined all-gather driver
(wgmath_amd/sharded.py) with an injected NumPy GEMM standing in for the HIP kernel (which needs a GPU) and check the
gathered cube views against the unsharded product."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from wgmath_amd.sharded import MShardPlan, ShardedGemm  # noqa: E402


def test_plan_views_cover_the_matrix_exactly_once():
    pl = MShardPlan(M=64, N=48, K=16, world=4, npanels=3)
    assert (pl.Mg, pl.np_) == (16, 16)
    seen = np.zeros(pl.gathered_elems(), np.int32)
    for r in range(pl.M):
        for c in range(pl.N):
            seen[pl.element_index(r, c)] += 1
    assert (seen == 1).all()
    # out_shape / cube_shape / element_index agree
    for p in range(pl.npanels):
        cube = pl.cube_shape(p)
        assert cube.size == (16, 16, 4) and cube.stride == 16 and cube.stride_mat == 256
        for g in range(pl.world):
            o = pl.out_shape(p, g)
            assert o.offset == cube.offset + g * cube.stride_mat  # == GpuCubeView::matrix(g).offset (tensor.rs:466-480)
            assert pl.element_index(g * 16 + 3, p * 16 + 5) == o.offset + 3 + 5 * o.stride
        start, n = pl.panel_range(p)
        assert start == cube.offset and n == 4 * 256
    b = pl.b_panel_shape(2)
    assert (b.size, b.stride, b.offset) == ((16, 16, 1), 16, 2 * 16 * 16)


def test_plan_rejects_unaligned_splits():
    with pytest.raises(ValueError):
        MShardPlan(M=72, N=48, K=16, world=4)      # 18 rows per rank: not vec4-aligned
    with pytest.raises(ValueError):
        MShardPlan(M=64, N=40, K=16, world=2, npanels=4)
    with pytest.raises(ValueError):
        MShardPlan(M=1 << 17, N=1 << 16, K=16, world=2)  # 2^33 elements


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, M, N, K, npanels, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # npanels may be an int (uniform panels) or a tuple of panel widths (ragged: what bench.py plans for a CU-masked compute stream)
        pl = MShardPlan(M, N, K, world, npanels) if isinstance(npanels, int) else MShardPlan(M, N, K, world, panel_cols=npanels)
        npanels = pl.npanels
        rng = np.random.default_rng(1234)  # same full A, B on every rank; each keeps its row block
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((K, N)).astype(np.float32)
        r0, nr = pl.a_rows(rank)
        a_g = np.ascontiguousarray(A[r0:r0 + nr].reshape(-1, order="F"))  # rank's own dense column-major tensor
        b_flat = B.reshape(-1, order="F")
        gathered = torch.full((pl.gathered_elems(),), float("nan"), dtype=torch.float32)
        g_np = gathered.numpy()  # shares memory

        def view_idx(s):
            i = np.arange(s.size[0])[:, None]
            j = np.arange(s.size[1])[None, :]
            return s.offset + i + j * s.stride

        def local_gemm(out_shape, a_shape, b_shape):  # NumPy stand-in for wg_gemm on the same three views
            a = a_g[view_idx(a_shape)]
            b = b_flat[view_idx(b_shape)]
            g_np[view_idx(out_shape)] = a @ b

        works = []

        def all_gather(start, count, rk):
            out = gathered[start:start + world * count]
            w = dist.all_gather_into_tensor(out, out[rk * count:(rk + 1) * count].clone(), async_op=True)
            works.append(w)
            return w

        ShardedGemm(pl, rank, local_gemm, all_gather, wait=lambda w: w.wait()).step()
        assert len(works) == npanels
        C = A @ B
        got = np.empty((M, N), np.float32)
        for p in range(npanels):  # read back through the cube views, matrix by matrix
            cube = pl.cube_shape(p)
            for g in range(world):
                nc, c0 = pl.cols_of(p), pl.col0_of(p)
                blk = g_np[cube.offset + g * cube.stride_mat + np.arange(pl.Mg)[:, None] + np.arange(nc)[None, :] * cube.stride]
                got[g * pl.Mg:(g + 1) * pl.Mg, c0:c0 + nc] = blk
        ok = bool(np.allclose(got, C, rtol=1e-5, atol=1e-5)) and not np.isnan(g_np).any()
        q.put((rank, ok, ""))
    except Exception as e:  # pragma: no cover
        q.put((rank, False, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("npanels", [1, 4, (8, 12, 8, 4)])
def test_sharded_gemm_world2_gloo(npanels):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, 64, 32, 24, npanels, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in results), results


def test_ragged_panel_plan():
    """Panels of unequal width (bench.py sizes them to whole rounds of the CUs the compute stream may use + a remainder):
    every element of C has exactly one home in the gathered buffer and the views tile it without gaps."""
    from wgmath_amd.sharded import MShardPlan
    pl = MShardPlan(64, 40, 8, 2, panel_cols=(16, 16, 8))
    assert pl.npanels == 3 and pl.Mg == 32
    assert [pl.cols_of(p) for p in range(3)] == [16, 16, 8] and [pl.col0_of(p) for p in range(3)] == [0, 16, 32]
    seen = np.zeros(pl.gathered_elems(), np.int32)
    for p in range(pl.npanels):
        start, n = pl.panel_range(p)
        assert n == pl.world * pl.slot_elems_of(p)
        for g in range(pl.world):
            o = pl.out_shape(p, g)
            assert o.offset == start + g * pl.slot_elems_of(p) and tuple(o.size[:2]) == (pl.Mg, pl.cols_of(p))
            idx = o.offset + np.arange(pl.Mg)[:, None] + np.arange(pl.cols_of(p))[None, :] * o.stride
            seen[idx.reshape(-1)] += 1
        b = pl.b_panel_shape(p)
        assert b.offset == pl.col0_of(p) * pl.K and tuple(b.size[:2]) == (pl.K, pl.cols_of(p))
    assert (seen == 1).all()
    for row, col in [(0, 0), (31, 15), (32, 16), (63, 39), (5, 33)]:
        p, j = pl.panel_of_col(col)
        g, i = divmod(row, pl.Mg)
        o = pl.out_shape(p, g)
        assert pl.element_index(row, col) == o.offset + i + j * o.stride
    with pytest.raises(ValueError):
        MShardPlan(64, 40, 8, 2, panel_cols=(16, 16, 4, 2))


def _tail_timeline(widths, ratio):
    """Exposed exchange time at the end of a step, in units of "Gemm time of one tile column": panel p's Gemm takes widths[p], its exchange
    ratio * widths[p]; the exchanges run one after the other on their own stream, each after its panel's Gemm."""
    t_gemm, t_exch = 0.0, 0.0
    for w in widths:
        t_gemm += w
        t_exch = max(t_exch, t_gemm) + ratio * w
    return t_exch - t_gemm


@pytest.mark.parametrize("N,main,ratio", [(32768, 2048, 0.72), (32768, 2048, 0.5), (8192, 2048, 0.72), (8448, 2048, 0.7), (32768 + 128, 2048, 0.72),
                                          (32768, 4096, 0.72), (16384, 1024, 0.72), (32768, 2304, 0.72), (4096, 2048, 0.72), (2048, 2048, 0.7), (256, 256, 0.7)])
def test_tapered_panel_plan(N, main, ratio):
    """The bench's N-panels with a tapered tail (wgmath_amd.sharded.tapered_panels): what the one-launch kernels take (equal panels of whole
    tiles, then <= 8 others, the last one the rest), widths never growing, and -- the point of it -- the exchange left exposed at the end of a
    step no longer that of a full panel."""
    from wgmath_amd.sharded import MShardPlan, tapered_panels
    w = tapered_panels(N, main, ratio)
    assert sum(w) == N and all(x > 0 for x in w)
    assert all(x % 256 == 0 for x in w[:-1]) and w[-1] % 4 == 0
    n_main = 0
    while n_main < len(w) and w[n_main] == main:
        n_main += 1
    tail = w[n_main:]
    assert len(tail) <= 8 and (n_main >= 1 or len(w) == 1)
    assert all(a >= b for a, b in zip(w[:-1], w[1:-1] + [w[-1] - N % 256])), f"widths must not grow: {w}"
    MShardPlan(8192, N, 64, 4, panel_cols=tuple(w))  # a valid ragged plan of the host-side planner too
    if N >= 4 * main:
        tiles = [x / 256 for x in w]
        uniform = [main / 256] * (N // main)
        assert tiles[-1] < 2.0, "the last panel is one tile column (plus the fraction of a ragged N)"
        assert _tail_timeline(tiles, ratio) <= 0.5 * _tail_timeline(uniform, ratio), (w, _tail_timeline(tiles, ratio), _tail_timeline(uniform, ratio))
    with pytest.raises(ValueError):
        tapered_panels(N, main + 4, ratio)
    with pytest.raises(ValueError):
        tapered_panels(N, main, 1.0)
