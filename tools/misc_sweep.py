"""Batched Gemm (nmats > 1), multi-RHS Gemv / GemvTr and Axpy against torch (bmm / matmul / add) on the same layouts: GPU time per call.
Usage (GPU box): python tools/misc_sweep.py   (torch first: one HIP runtime per process)"""
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


def timed(fn, reps):
    def run(n):
        enc = dev.create_command_encoder()
        p = enc.compute_pass("t", None)
        for _ in range(n):
            fn(p)
        p.end()
        gpu.queue().submit([enc.finish()])
        gpu.sync()
    run(3)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        run(reps)
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


def ttimed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


gemm = wg.Gemm.from_device(dev)
for dt, td in ((np.float16, torch.float16), (np.float32, torch.float32)):
    for (M, N, K, mats) in [] if os.environ.get("MISC_GEMV_ONLY") else [(128, 128, 128, 64), (256, 256, 256, 64), (512, 512, 512, 16), (1024, 1024, 1024, 8), (2048, 2048, 2048, 4), (64, 64, 64, 1024), (128, 128, 4096, 32),
                            (1024, 1024, 128, 32), (4096, 4096, 4096, 2)]:
        a = device_random(wg, gpu, (M, K, mats), dt, 1)
        b = device_random(wg, gpu, (K, N, mats), dt, 2)
        c = wg.TensorBuilder.tensor((M, N, mats), S.STORAGE).build(dev, dt)
        reps = max(10, min(200, int(0.1 / (2.0 * M * N * K * mats / (5e14 if dt == np.float16 else 8e13) + 5e-6))))
        to = timed(lambda p: gemm.dispatch(dev, shapes, p, c, a, b), reps)
        bt = (torch.rand(mats, N, K, device="cuda") * 2 - 1).to(td)
        at = (torch.rand(mats, K, M, device="cuda") * 2 - 1).to(td)
        ct = torch.empty(mats, N, M, device="cuda", dtype=td)
        tv = ttimed(lambda: torch.bmm(bt, at, out=ct), reps)
        f = 2.0 * M * N * K * mats / 1e12
        print(f"gemm batched {np.dtype(dt).name} {M}x{N}x{K} x{mats}: ours {to*1e6:8.1f} us {f/to:7.1f} TF | vendor {tv*1e6:8.1f} us {f/tv:7.1f} TF | ratio {to/tv:5.2f}{'  <-- behind' if to > 1.1 * tv else ''}", flush=True)
        del a, b, c, bt, at, ct

gemv = wg.Gemv.from_device(dev)
for dt, td in ((np.float32, torch.float32), (np.float16, torch.float16)):
    GEMV_SHAPES = [(4096, 4096, 2), (4096, 4096, 4), (4096, 4096, 8), (4096, 65536, 8), (65536, 4096, 8), (4096, 11008, 4), (11008, 4096, 4), (8192, 8192, 3)]
    if os.environ.get("MISC_GEMV_SHAPES"):  # "RxCxNRHS ..." instead (e.g. single-vector shapes: 4096x4096x1)
        GEMV_SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["MISC_GEMV_SHAPES"].split()]
    for (R, C, nrhs) in GEMV_SHAPES:
        for tr in (False, True):
            m = device_random(wg, gpu, (R, C), dt, 1)
            nin, nout = (R, C) if tr else (C, R)
            x = device_random(wg, gpu, (nin, nrhs), dt, 2)
            y = wg.TensorBuilder.matrix(nout, nrhs, S.STORAGE).build(dev, dt)
            variant = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv
            to = timed(lambda p: gemv.dispatch_generic(dev, shapes, p, y, m, x, variant), 100)
            mt = (torch.rand(C, R, device="cuda") * 2 - 1).to(td)  # column-major R x C
            xt = (torch.rand(nrhs, nin, device="cuda") * 2 - 1).to(td)
            yt = torch.empty(nrhs, nout, device="cuda", dtype=td)
            # y^T (nrhs x nout) = x^T (nrhs x nin) op(M)^T:  N: x^T (nrhs x C) @ M^T (C x R) = mt;  T: x^T (nrhs x R) @ M (R x C) = mt.t()
            tv = ttimed(lambda: torch.matmul(xt, mt.t() if tr else mt, out=yt), 100)
            byts = np.dtype(dt).itemsize * (R * C + (nin + nout) * nrhs)
            print(f"{'gemv_tr' if tr else 'gemv   '} {np.dtype(dt).name} {R}x{C} rhs={nrhs}: ours {to*1e6:8.1f} us {byts/to/1e9:6.0f} GB/s | vendor {tv*1e6:8.1f} us {byts/tv/1e9:6.0f} GB/s | ratio {to/tv:5.2f}{'  <-- behind' if to > 1.1 * tv else ''}", flush=True)
            del m, x, y, mt, xt, yt
