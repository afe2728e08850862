#!/usr/bin/env python3
"""f16 Gemm on outputs of one to a few rounds of 256 x 256 tiles: the launcher's choice, both tile families forced (wg_ctx_set_tuning), and hipBLASLt
through torch.matmul when torch is importable. usage: python tools/midsize_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
try:
    import torch
except Exception:
    torch = None
import wgmath_amd as wg
from bench import device_random
gpu = wg.GpuInstance.new(0); dev, shapes = gpu.device(), wg.ViewShapeBuffers(); S = wg.BufferUsages
gemm = wg.Gemm.from_device(dev)
def ours(M, N, K, tile):
    gpu.set_tuning("f16_tile", int(tile) if tile else 0)
    a = device_random(wg, gpu, (M, K), np.float16, 1); b = device_random(wg, gpu, (K, N), np.float16, 2)
    c = wg.TensorBuilder.matrix(M, N, S.STORAGE).build(dev, np.float16)
    enc = dev.create_command_encoder(); p = enc.compute_pass("t", None)
    for _ in range(5): gemm.dispatch(dev, shapes, p, c, a, b)
    gpu.sync(); n = max(20, int(0.3 / (2.0 * M * N * K / 1.2e15))); t0 = time.perf_counter()
    for _ in range(n): gemm.dispatch(dev, shapes, p, c, a, b)
    gpu.sync(); dt = (time.perf_counter() - t0) / n
    gpu.set_tuning("f16_tile", 0)
    return dt
def vendor(M, N, K):
    if torch is None: return float("nan")
    a = (torch.rand(K, M, device="cuda") * 2 - 1).half().t(); b = (torch.rand(N, K, device="cuda") * 2 - 1).half().t()  # col-major A (m-contiguous), B k-contiguous
    c = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(5): torch.matmul(a, b, out=c)
    torch.cuda.synchronize(); n = max(20, int(0.3 / (2.0 * M * N * K / 1.2e15))); t0 = time.perf_counter()
    for _ in range(n): torch.matmul(a, b, out=c)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for (M, N, K) in [(4096, 4096, 4096), (4096, 4096, 8192), (6144, 6144, 6144), (5120, 5120, 5120), (3072, 3072, 3072), (4096, 8192, 4096)]:
    f = 2.0 * M * N * K / 1e12
    ta, t2, t1, tv = ours(M, N, K, None), ours(M, N, K, "256"), ours(M, N, K, "128"), vendor(M, N, K)
    print(f"{M}x{N}x{K}: auto {ta*1e6:8.1f} us {f/ta:7.1f} TF | 256-tile {f/t2:7.1f} | 128-tile {f/t1:7.1f} | vendor {f/tv:7.1f}", flush=True)
