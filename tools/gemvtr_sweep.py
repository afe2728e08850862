"""GemvTr over a sweep of shapes (the stored matrix is R x C column-major, out = m^T v has C entries), f32 and f16: GPU time per dispatch
from 200 back-to-back eager launches. Usage (GPU box): WGEBRA_HIP_LIB=... python tools/gemvtr_sweep.py [RxC ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench as B  # noqa: E402
import wgmath_amd as wg  # noqa: E402

gpu = wg.GpuInstance.new(0)
S = wg.BufferUsages
SHAPES = [(65536, 4096), (32768, 4096), (16384, 4096), (8192, 4096), (4096, 4096), (8192, 8192), (16384, 8192), (32768, 8192), (65536, 1024), (65536, 256),
          (262144, 256), (1048576, 64), (11008, 4096), (4096, 11008), (4096, 32000), (2048, 2048), (131072, 2048)]
TR = os.environ.get("SWEEP_N") != "1"  # SWEEP_N=1: Gemv (out = m v, R entries) instead of GemvTr
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for dt in (np.float32, np.float16):
    for R, C in SHAPES:
        A = B.device_random(wg, gpu, (R, C), dt, 1)
        x = B.device_random(wg, gpu, (R if TR else C,), dt, 2)
        y = wg.TensorBuilder.vector(C if TR else R, S.STORAGE | S.COPY_SRC).build(gpu.device(), dt)
        gemv = wg.Gemv.from_device(gpu.device())
        shapes = wg.ViewShapeBuffers()

        def run(n):
            enc = gpu.device().create_command_encoder()
            p = enc.compute_pass("gemv", None)
            for _ in range(n):
                gemv.dispatch_generic(gpu.device(), shapes, p, y.as_embedded_view(2), A.as_embedded_view(3), x.as_embedded_view(2), wg.GemvVariant.GemvTr if TR else wg.GemvVariant.Gemv)
            p.end()
            gpu.queue().submit([enc.finish()])
            gpu.sync()
        run(10)
        if R * C <= 4096 * 4096:  # numbers too: against f64 on the host
            a64 = A.slow_read(gpu).astype(np.float64).reshape(C, R)
            ref = a64 @ x.slow_read(gpu).astype(np.float64) if TR else a64.T @ x.slow_read(gpu).astype(np.float64)
            got = y.slow_read(gpu).astype(np.float64)
            err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
            assert err < (2e-3 if dt == np.float16 else 1e-5), (R, C, dt, err)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            run(200)
            best = min(best, (time.perf_counter() - t0) / 200)
        byts = np.dtype(dt).itemsize * (R * C + R + C)
        print(f"{'gemv_tr' if TR else 'gemv'} {np.dtype(dt).name} {R}x{C}: {best*1e6:8.1f} us {byts/best/1e9:6.0f} GB/s", flush=True)
        del A, x, y
