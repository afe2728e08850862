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
m1 = wg.TensorBuilder.tensor((K, M, 1), S_STORAGE).build_init(dev, a, np.float16)
m2 = wg.TensorBuilder.tensor((K, N, 1), S_STORAGE).build_init(dev, b, np.float16)
gpu.set_tuning("f16_tile", 256)


def run(cont):
    gpu.set_tuning("f16_cont", cont)
    out = wg.TensorBuilder.tensor((M, N, 1), S_STORAGE).build_init(dev, np.full(M * N, np.nan, np.float16), np.float16)
    enc = dev.create_command_encoder()
    p = enc.compute_pass("t", None)
    gemm.dispatch_generic(dev, vs, p, out, m1, m2, wg.GemmVariant.GemmTr)
    p.end()
    gpu.queue().submit([enc.finish()])
    gpu.sync()
    return out.read(dev).view(np.uint16).reshape(N, M).copy()


ref = run(0)
for it in range(int(os.environ.get("ITERS", "6"))):
    got = run(1)
    d = got != ref
    n = int(d.sum())
    print(f"run {it}: {n} differing, nan in continuous {int(np.isnan(got.view(np.float16)).sum())}", flush=True)
    if n:
        t = d.reshape(N // 256, 256, M // 256, 256).sum(axis=(1, 3))  # [tn][tm]
        bad = np.argwhere(t)
        print("  tiles (tn, tm, count):", [(int(x), int(y), int(t[x, y])) for x, y in bad[:24]], "..." if len(bad) > 24 else "")
        cols = d.reshape(N // 256, 256, M).sum(axis=(0, 2))
        rows = d.reshape(N, M // 256, 256).sum(axis=(0, 1))
        print("  by column in tile (nonzero):", {int(i): int(c) for i, c in enumerate(cols) if c}.__repr__()[:600])
        print("  by row in tile (nonzero):", {int(i): int(c) for i, c in enumerate(rows) if c}.__repr__()[:600])
        y, x = np.argwhere(d)[0]
        print("  first:", (int(y), int(x)), "ref", ref.view(np.float16)[y, x], "got", got.view(np.float16)[y, x])
        idx = np.argwhere(d)[:2000]
        rel = np.abs(got.view(np.float16)[idx[:, 0], idx[:, 1]].astype(np.float64) - ref.view(np.float16)[idx[:, 0], idx[:, 1]].astype(np.float64))
        print("  |diff| median / max over the first 2000:", float(np.median(rel)), float(rel.max()))
