"""GEMV / GemvTr at LLM-decode-like sizes: this library (eager and as a recorded command buffer) vs torch.mv (rocBLAS), GPU-side time
per call from back-to-back launches. Usage: python tools/gemv_probe.py  (GPU box). torch first: one HIP runtime per process."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

import bench as B  # noqa: E402
import wgmath_amd as wg  # noqa: E402

gpu = wg.GpuInstance.new(0)
S = wg.BufferUsages
SHAPES = [(4096, 4096), (4096, 11008), (11008, 4096), (32000, 4096), (8192, 8192), (16384, 1024), (1024, 16384), (2048, 2048), (512, 512), (256, 65536), (65536, 256)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for R, C in SHAPES:
    for tr in (False, True):
        # y = A x (A: R x C column-major) or y = A^T x (A stored C x R... keep the SAME matrix: GemvTr of an R x C matrix gives C outputs)
        A = B.device_random(wg, gpu, (R, C), np.float32, 1)
        nin, nout = (R, C) if tr else (C, R)
        x = B.device_random(wg, gpu, (nin,), np.float32, 2)
        y = wg.TensorBuilder.vector(nout, S.STORAGE | S.COPY_SRC).build(gpu.device(), np.float32)
        gemv = wg.Gemv.from_device(gpu.device())
        shapes = wg.ViewShapeBuffers()
        variant = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv

        def enc_n(n, record):
            enc = gpu.device().create_command_encoder(record=record)
            p = enc.compute_pass("gemv", None)
            for _ in range(n):
                gemv.dispatch_generic(gpu.device(), shapes, p, y.as_embedded_view(2), A.as_embedded_view(3), x.as_embedded_view(2), variant)
            p.end()
            return enc.finish()
        gpu.queue().submit([enc_n(5, False)])
        gpu.sync()
        reps = 200
        t0 = time.perf_counter()
        gpu.queue().submit([enc_n(reps, False)])
        gpu.sync()
        eager = (time.perf_counter() - t0) / reps
        cb = enc_n(50, True)
        gpu.queue().submit([cb]); gpu.sync()
        t0 = time.perf_counter()
        for _ in range(4):
            gpu.queue().submit([cb])
        gpu.sync()
        graph = (time.perf_counter() - t0) / 200
        # vendor: same memory (column-major R x C == row-major C x R)
        a_t = (torch.rand(C, R, device="cuda") * 2 - 1)
        xv = torch.rand(nin, device="cuda")
        yv = torch.empty(nout, device="cuda")
        m = a_t if tr else a_t.t()
        for _ in range(5):
            torch.mv(m, xv, out=yv)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            torch.mv(m, xv, out=yv)
        torch.cuda.synchronize()
        vend = (time.perf_counter() - t0) / reps
        byts = 4.0 * (R * C + R + C)
        print(f"{'gemv_tr' if tr else 'gemv   '} {R}x{C}: eager {eager*1e6:7.1f} us {byts/eager/1e9:6.0f} GB/s | recorded {graph*1e6:7.1f} us {byts/graph/1e9:6.0f} GB/s | vendor {vend*1e6:7.1f} us {byts/vend/1e9:6.0f} GB/s", flush=True)
        del A, x, y, a_t
