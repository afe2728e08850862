#!/usr/bin/env python3
"""Shape sweep: sustained throughput of Gemm / Gemv over square and tall-skinny shapes (not part of the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wgmath_amd as wg
sys.argv = sys.argv[:1] + sys.argv[1:]
from bench import device_random

gpu = wg.GpuInstance.new(0)
dev, shapes = gpu.device(), wg.ViewShapeBuffers()
S = wg.BufferUsages
enc = dev.create_command_encoder(); p = enc.compute_pass("sweep", None)

def timeit(fn, min_s=0.25):
    for _ in range(3): fn()
    gpu.sync(); t0 = time.perf_counter(); fn(); gpu.sync(); one = max(time.perf_counter() - t0, 1e-6)
    n = int(min(max(5, min_s / one), 5000))
    ts = wg.GpuTimestamps.new(dev, 2); ts.write(dev)
    for _ in range(n): fn()
    ts.write(dev); t = ts.wait_for_results_ms()
    return (t[1] - t[0]) / n * 1e-3

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "gemv"):
    gemv = wg.Gemv.from_device(dev)
    print("GEMV f32:  R x C      N: GB/s (%HBM)     T: GB/s (%HBM)")
    for R, C in [(1024, 1024), (4096, 4096), (8192, 8192), (16384, 16384), (32768, 8192), (8192, 32768), (4096, 65536), (65536, 4096),
                 (1 << 20, 64), (64, 1 << 20), (1 << 18, 1024), (1024, 1 << 18), (512, 512)]:
        m = device_random(wg, gpu, (R, C), np.float32, 1)
        res = []
        for tr in (False, True):
            vlen, olen = (R, C) if tr else (C, R)
            v = device_random(wg, gpu, (vlen,), np.float32, 2)
            o = wg.TensorBuilder.vector(olen, S.STORAGE).build(dev, np.float32)
            var = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv
            dt = timeit(lambda: gemv.dispatch_generic(dev, shapes, p, o, m, v, var))
            gbs = 4.0 * (R * C + R + C) / dt / 1e9
            res.append(f"{gbs:8.0f} ({gbs / 80:4.1f}%) {dt*1e6:8.1f}us")
        print(f"  {R:8d} x {C:8d}   {res[0]}   {res[1]}")
        del m
if which in ("all", "gemm"):
    gemm = wg.Gemm.from_device(dev)
    for dtype, peak in ((np.float32, 157.3), (np.float16, 2500.0)):
        print(f"GEMM {np.dtype(dtype).name}:  M x K x N      NN: TF (%peak)      TN: TF (%peak)")
        for M, K, N in [(1024, 1024, 1024), (2048, 2048, 2048), (4096, 4096, 4096), (8192, 8192, 8192), (65536, 1024, 1024), (1 << 20, 256, 256),
                        (1024, 65536, 1024), (256, 256, 1 << 20), (8192, 512, 8192), (4104, 4096, 4104), (16384, 16384, 256)]:
            if dtype == np.float32 and M * K * N > 4096 ** 3 * 4: continue
            res = []
            for tr in (False, True):
                a = device_random(wg, gpu, (K, M) if tr else (M, K), dtype, 3)
                b = device_random(wg, gpu, (K, N), dtype, 4)
                c = wg.TensorBuilder.matrix(M, N, S.STORAGE).build(dev, dtype)
                var = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
                dt = timeit(lambda: gemm.dispatch_generic(dev, shapes, p, c, a, b, var))
                tf = 2.0 * M * N * K / dt / 1e12
                res.append(f"{tf:8.1f} ({tf / peak * 100:4.1f}%) {dt*1e6:9.1f}us")
                del a, b, c
            print(f"  {M:8d} x {K:6d} x {N:8d}   {res[0]}   {res[1]}")
