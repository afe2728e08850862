"""Tall-skinny GEMM (M >> N): this library vs the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) on the same shapes.
Usage: python tools/tall_skinny_probe.py  (GPU box). torch is imported first: one HIP runtime per process."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench as B  # noqa: E402
import wgmath_amd as wg  # noqa: E402

gpu = wg.GpuInstance.new(0)
SHAPES = [("f16", 131072, 1024, 8192), ("f16", 262144, 256, 4096), ("f16", 65536, 2048, 8192), ("f16", 524288, 256, 1024),
          ("f16", 1048576, 256, 256), ("f32", 65536, 512, 4096), ("f32", 262144, 128, 2048), ("f32", 131072, 256, 4096)]
if len(sys.argv) > 1:  # e.g. f16:2048x2048x2048 f32:1024x1024x1024
    SHAPES = [(a.split(":")[0],) + tuple(int(x) for x in a.split(":")[1].split("x")) for a in sys.argv[1:]]
for dt, M, N, K in SHAPES:
    npdt = np.float16 if dt == "f16" else np.float32
    TR = os.environ.get("TR") == "1"  # GemmTr: m1 stored K x M
    A = B.device_random(wg, gpu, (K, M) if TR else (M, K), npdt, 1)
    Bm = B.device_random(wg, gpu, (K, N), npdt, 2)
    S = wg.BufferUsages
    C = wg.TensorBuilder.matrix(M, N, S.STORAGE | S.COPY_SRC).build(gpu.device(), npdt)
    gemm = wg.Gemm.from_device(gpu.device())
    shapes = wg.ViewShapeBuffers()
    enc = gpu.device().create_command_encoder()
    p = enc.compute_pass("ts", None)
    for tr in (False,):
        def go():
            gemm.dispatch_generic(gpu.device(), shapes, p, C.as_embedded_view(3), A.as_embedded_view(3), Bm.as_embedded_view(3),
                                  wg.GemmVariant.GemmTr if TR else wg.GemmVariant.Gemm)
        for _ in range(5):
            go()
        gpu.sync()
        reps = 200 if M * N * K < (1 << 36) else 30
        t0 = time.perf_counter()
        for _ in range(reps):
            go()
        gpu.sync()
        ours = (time.perf_counter() - t0) / reps
    tdt = torch.float16 if dt == "f16" else torch.float32
    # column-major M x K == row-major K x M transposed: the same memory layout the library sees
    a = (torch.rand(K, M, device="cuda", dtype=torch.float32) * 2 - 1).to(tdt).t()
    b = (torch.rand(N, K, device="cuda", dtype=torch.float32) * 2 - 1).to(tdt).t()
    c = torch.empty(N, M, device="cuda", dtype=tdt).t()
    for _ in range(5):
        torch.matmul(a, b, out=c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        torch.matmul(a, b, out=c)
    torch.cuda.synchronize()
    vend = (time.perf_counter() - t0) / reps
    fl = 2.0 * M * N * K
    byts = (M * K + K * N + M * N) * np.dtype(npdt).itemsize
    print(f"{dt} {M}x{N}x{K}: ours {ours*1e3:.3f} ms {fl/ours/1e12:.1f} TF ({byts/ours/1e9:.0f} GB/s)  vendor {vend*1e3:.3f} ms {fl/vend/1e12:.1f} TF", flush=True)
    del A, Bm, C, a, b, c
