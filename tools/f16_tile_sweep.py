#!/usr/bin/env python3
"""f16 Gemm / GemmTr with every tile family forced (WG_TUNE_F16_TILE: 128 x 128, 256 x 256 per-tile / continuous walk, 256 x 128 pairs) next to the launcher's own choice
and the vendor library (torch.matmul on the same layout): GPU time per call from back-to-back launches. Where the launcher's column is not the smallest, its model is off.
Usage (GPU box): python tools/f16_tile_sweep.py [MxNxK[xB] ...]   (torch first: one HIP runtime per process)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import wgmath_amd as wg  # noqa: E402
from bench import device_random  # noqa: E402

gpu = wg.GpuInstance.new(0)
dev, shapes = gpu.device(), wg.ViewShapeBuffers()
S = wg.BufferUsages
gemm = wg.Gemm.from_device(dev)
SHAPES = [(4096, 4096, 256), (4096, 4096, 512), (4096, 4096, 1024), (4096, 4096, 2048), (6144, 6144, 256), (6144, 6144, 1024), (3072, 3072, 1024), (3072, 3072, 3072), (2048, 8192, 512),
          (5120, 5120, 512), (2560, 2560, 2560), (1024, 1024, 1024, 8), (2048, 2048, 1024, 4), (4096, 2048, 4096)]
if sys.argv[1:]:
    SHAPES = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
SHAPES = [s if len(s) == 4 else tuple(s) + (1,) for s in SHAPES]
KNOBS = [("128", 128, -1), ("256 per-tile", 256, 0), ("256 walk", 256, 1), ("256x128", 256128, -1), ("auto", 0, -1)]


def timed(fn, flops):
    def run(n):
        enc = dev.create_command_encoder()
        p = enc.compute_pass("t", None)
        for _ in range(n):
            fn(p)
        p.end()
        gpu.queue().submit([enc.finish()])
        gpu.sync()
    run(3)
    n = max(10, min(300, int(0.1 / (flops / 1.0e15 + 5e-6))))
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        run(n)
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6


def vendor(M, N, K, B, tr):
    bt = (torch.rand(B, N, K, device="cuda") * 2 - 1).to(torch.float16)
    at = (torch.rand(B, M, K, device="cuda") * 2 - 1).to(torch.float16).transpose(1, 2) if tr else (torch.rand(B, K, M, device="cuda") * 2 - 1).to(torch.float16)
    ct = torch.empty(B, N, M, device="cuda", dtype=torch.float16)
    if B == 1:
        bt, at, ct = bt[0], at[0], ct[0]
    for _ in range(3):
        torch.matmul(bt, at, out=ct)
    torch.cuda.synchronize()
    n = max(10, min(300, int(0.1 / (2.0 * M * N * K * B / 1.0e15 + 5e-6))))
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(n):
            torch.matmul(bt, at, out=ct)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6


print("# us per call: " + " | ".join(k[0] for k in KNOBS) + " | vendor ; auto / best of ours, auto / vendor")
for M, N, K, B in SHAPES:
    for tr in (False, True):
        a = device_random(wg, gpu, (K, M, B) if tr else (M, K, B), np.float16, 1)
        b = device_random(wg, gpu, (K, N, B), np.float16, 2)
        c = wg.TensorBuilder.tensor((M, N, B), S.STORAGE).build(dev, np.float16)
        variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
        row = []
        for _, tile, cont in KNOBS:
            gpu.set_tuning("f16_tile", tile)
            gpu.set_tuning("f16_cont", cont)
            try:
                row.append(timed(lambda p: gemm.dispatch_generic(dev, shapes, p, c, a, b, variant), 2.0 * M * N * K * B))
            except Exception:  # a forced family that does not take the shape
                row.append(float("nan"))
        gpu.set_tuning("f16_tile", 0)
        gpu.set_tuning("f16_cont", -1)
        v = vendor(M, N, K, B, tr)
        best = np.nanmin(row[:-1])
        flag = "  <-- model off" if row[-1] > 1.05 * best else ""
        print(f"f16 {'gemm_tr' if tr else 'gemm   '} {M}x{N}x{K}x{B}: " + " | ".join(f"{t:7.1f}" for t in row) + f" | vendor {v:7.1f} ; {row[-1] / best:5.2f} {row[-1] / v:5.2f}{flag}", flush=True)
