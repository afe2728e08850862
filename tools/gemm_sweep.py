"""Gemm / GemmTr over a sweep of shapes (squares, LLM-like projections, few rows / few columns, ragged), f16 and f32, against hipBLASLt / rocBLAS through
torch.matmul on the SAME memory layout (column-major C = A B is row-major C^T = B^T A^T). GPU time per call from back-to-back launches.
Usage (GPU box): python tools/gemm_sweep.py [rm] [f16|f32] [MxNxK[xB] ...]   (B: a batch of B matrices; torch first: one HIP runtime per process)
`rm`: the ROW-major operator surface instead (wg_gemm_rm on row-major views = torch's own layout: C = A @ B and C = A^T @ B on contiguous tensors)."""
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
SHAPES = [(512, 512, 512), (1024, 1024, 1024), (1536, 1536, 1536), (2048, 2048, 2048), (3072, 3072, 3072), (4096, 4096, 4096), (5120, 5120, 5120), (6144, 6144, 6144),
          (8192, 8192, 8192), (8192, 4096, 4096), (4096, 11008, 4096), (4096, 4096, 11008), (2048, 14336, 4096), (8192, 8192, 1024), (1024, 8192, 8192),
          (8192, 1024, 8192), (16, 4096, 4096), (64, 4096, 4096), (128, 11008, 4096), (256, 4096, 11008), (4096, 16, 4096), (4096, 64, 4096), (4096, 128, 4096),
          (4104, 4104, 4104), (8200, 8200, 8200), (1000, 1000, 1000), (32768, 1024, 1024), (1024, 32768, 1024), (1024, 1024, 32768)]
args = sys.argv[1:]
ROW_MAJOR = bool(args) and args[0] == "rm"
if ROW_MAJOR:
    args = args[1:]
    gemm = wg.Gemm.from_device(dev, wg.row_major_shader_defs())
dts = [np.float16, np.float32]
if args and args[0] in ("f16", "f32"):
    dts = [np.float16 if args[0] == "f16" else np.float32]
    args = args[1:]
if args:
    SHAPES = [tuple(int(x) for x in a.split("x")) for a in args]
SHAPES = [s if len(s) == 4 else tuple(s) + (1,) for s in SHAPES]


def reps_for(M, N, K, dt):
    rate = 1.0e15 if dt == np.float16 else 1.0e14
    return max(10, min(300, int(0.15 / (2.0 * M * N * K / rate + 5e-6))))


def ours(M, N, K, dt, tr, B=1):
    a = device_random(wg, gpu, (K, M, B) if tr else (M, K, B), dt, 1)
    b = device_random(wg, gpu, (K, N, B), dt, 2)
    c = wg.TensorBuilder.tensor((M, N, B), S.STORAGE).build(dev, dt)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm

    def run(n):
        enc = dev.create_command_encoder()
        p = enc.compute_pass("t", None)
        for _ in range(n):
            gemm.dispatch_generic(dev, shapes, p, c, a, b, variant)
        p.end()
        gpu.queue().submit([enc.finish()])
        gpu.sync()
    run(3)
    n = reps_for(M, N, K * B, dt)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        run(n)
        best = min(best, (time.perf_counter() - t0) / n)
    return best


def ours_rm(M, N, K, dt, tr, B=1):
    """Row-major views over plain buffers: m1 (M x K, or K x M for GemmTr), m2 (K x N), out (M x N); index = mat * rows * cols + i * cols + j."""
    def rm(t, rows, cols):
        return wg.GpuTensorView(wg.ViewShape([rows, cols, B], cols, rows * cols, 0), t, 3)
    ta, tb = device_random(wg, gpu, (M * K * B,), dt, 1), device_random(wg, gpu, (K * N * B,), dt, 2)
    tc = wg.TensorBuilder.tensor((M * N * B,), S.STORAGE).build(dev, dt)
    va, vb, vc = (rm(ta, K, M) if tr else rm(ta, M, K)), rm(tb, K, N), rm(tc, M, N)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm

    def run(n):
        enc = dev.create_command_encoder()
        p = enc.compute_pass("t", None)
        for _ in range(n):
            gemm.dispatch_generic(dev, shapes, p, vc, va, vb, variant)
        p.end()
        gpu.queue().submit([enc.finish()])
        gpu.sync()
    run(3)
    n = reps_for(M, N, K * B, dt)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        run(n)
        best = min(best, (time.perf_counter() - t0) / n)
    return best


def vendor_rm(M, N, K, dt, tr, B=1):
    td = torch.float16 if dt == np.float16 else torch.float32
    a = (torch.rand((B, K, M) if tr else (B, M, K), device="cuda") * 2 - 1).to(td)
    b = (torch.rand(B, K, N, device="cuda") * 2 - 1).to(td)
    c = torch.empty(B, M, N, device="cuda", dtype=td)
    if B == 1:
        a, b, c = a[0], b[0], c[0]
    aa = a.transpose(-1, -2) if tr else a
    for _ in range(3):
        torch.matmul(aa, b, out=c)
    torch.cuda.synchronize()
    n = reps_for(M, N, K * B, dt)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(n):
            torch.matmul(aa, b, out=c)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best


def vendor(M, N, K, dt, tr, B=1):
    td = torch.float16 if dt == np.float16 else torch.float32
    bt = (torch.rand(B, N, K, device="cuda") * 2 - 1).to(td)                    # B^T row-major = B column-major (k-contiguous)
    at = (torch.rand(B, M, K, device="cuda") * 2 - 1).to(td).transpose(1, 2) if tr else (torch.rand(B, K, M, device="cuda") * 2 - 1).to(td)  # A^T as (K, M): tr -> k-contiguous storage
    ct = torch.empty(B, N, M, device="cuda", dtype=td)
    if B == 1:
        bt, at, ct = bt[0], at[0], ct[0]
    for _ in range(3):
        torch.matmul(bt, at, out=ct)
    torch.cuda.synchronize()
    n = reps_for(M, N, K * B, dt)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(n):
            torch.matmul(bt, at, out=ct)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best


for dt in dts:
    for tr in (False, True):
        for (M, N, K, B) in SHAPES:
            f = 2.0 * M * N * K * B / 1e12
            to, tv = (ours_rm(M, N, K, dt, tr, B), vendor_rm(M, N, K, dt, tr, B)) if ROW_MAJOR else (ours(M, N, K, dt, tr, B), vendor(M, N, K, dt, tr, B))
            flag = "  <-- behind" if to > 1.10 * tv else ""
            print(f"{np.dtype(dt).name} {'rm ' if ROW_MAJOR else ''}{'gemm_tr' if tr else 'gemm   '} {M}x{N}x{K}{'' if B == 1 else 'x' + str(B)}: ours {to*1e6:9.1f} us {f/to:7.1f} TF | vendor {tv*1e6:9.1f} us {f/tv:7.1f} TF | ours/vendor time {to/tv:5.2f}{flag}", flush=True)
