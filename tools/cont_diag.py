#!/usr/bin/env python3
"""Where the continuous f16 GemmTr walk differs from the per-tile launch (debugging aid): per 256 x 256 tile, and by position inside the tile."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import wgmath_amd as wg  # noqa: E402

M, N, K = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8192x8192x256").split("x"))
S_STORAGE = 128 | 4 | 8
gpu = wg.GpuInstance.new(0)
dev, vs = gpu.device(), wg.ViewShapeBuffers()
gemm = wg.Gemm.from_device(dev)
rng = np.random.default_rng(M + K + N)
a = (rng.random(M * K, dtype=np.float32) * 2 - 1).astype(np.float16)
b = (rng.random(K * N, dtype=np.float32) * 2 - 1).astype(np.float16)
tr = os.environ.get("CONT_VARIANT", "gemmtr") == "gemmtr"
m1 = wg.TensorBuilder.tensor((K, M, 1) if tr else (M, K, 1), S_STORAGE).build_init(dev, a, np.float16)
m2 = wg.TensorBuilder.tensor((K, N, 1), S_STORAGE).build_init(dev, b, np.float16)
gpu.set_tuning("f16_tile", 256)


def run(cont):
    gpu.set_tuning("f16_cont", cont)
    out = wg.TensorBuilder.tensor((M, N, 1), S_STORAGE).build_init(dev, np.full(M * N, np.nan, np.float16), np.float16)
    enc = dev.create_command_encoder()
    p = enc.compute_pass("t", None)
    gemm.dispatch_generic(dev, vs, p, out, m1, m2, wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm)
    p.end()
    gpu.queue().submit([enc.finish()])
    gpu.sync()
    return out.read(dev).view(np.uint16).reshape(N, M).copy()


ref = run(0)
ref2 = run(0)
print("per-tile launch twice: differing", int((ref != ref2).sum()), flush=True)
A = (a.reshape(M, K) if tr else a.reshape(K, M).T).astype(np.float32)  # row m of op(A)
Bm = b.reshape(N, K).astype(np.float32)
for it in range(int(os.environ.get("ITERS", "6"))):
    got = run(1)
    d = got != ref
    n = int(d.sum())
    print(f"run {it}: {n} differing, nan in continuous {int(np.isnan(got.view(np.float16)).sum())}", flush=True)
    if n:
        idx_all = np.argwhere(d)
        tiles = {}
        for yy, xx in idx_all[:100000]:
            tiles[(int(yy) // 256, int(xx) // 256)] = tiles.get((int(yy) // 256, int(xx) // 256), 0) + 1
        print("  tiles (tn, tm): count:", dict(list(tiles.items())[:24]))
        print("  columns in tile:", sorted(set(int(v) % 256 for v in idx_all[:100000, 0]))[:40], " rows in tile:", sorted(set(int(v) % 256 for v in idx_all[:100000, 1]))[:40])
        for yy, xx in idx_all[:4]:
            exact = float(A[xx].astype(np.float64) @ Bm[yy].astype(np.float64))
            print(f"  (col {yy}, row {xx}): per-tile {ref.view(np.float16)[yy, xx]}  continuous {got.view(np.float16)[yy, xx]}  f64 {exact:.6f}")
        y, x = idx_all[0]
        print("  first:", (int(y), int(x)), "ref", ref.view(np.float16)[y, x], "got", got.view(np.float16)[y, x])
        idx = np.argwhere(d)[:2000]
        rel = np.abs(got.view(np.float16)[idx[:, 0], idx[:, 1]].astype(np.float64) - ref.view(np.float16)[idx[:, 0], idx[:, 1]].astype(np.float64))
        print("  |diff| median / max over the first 2000:", float(np.median(rel)), float(rel.max()))
