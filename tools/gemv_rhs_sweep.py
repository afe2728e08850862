"""Gemv / GemvTr with 1 .. 8 right-hand sides over a few shapes (the stored matrix is R x C column-major), f32 and f16: GPU time per dispatch from
200 back-to-back eager launches and the HBM rate of the algorithmic bytes. Usage (GPU box): [WGEBRA_HIP_LIB=...] python tools/gemv_rhs_sweep.py [RxCxNRHS ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench as B  # noqa: E402
import wgmath_amd as wg  # noqa: E402

gpu = wg.GpuInstance.new(0)
S = wg.BufferUsages
CASES = [(4096, 65536, 1), (4096, 65536, 2), (4096, 65536, 4), (4096, 65536, 8), (16384, 16384, 1), (16384, 16384, 8), (8192, 8192, 1), (8192, 8192, 2), (8192, 8192, 8),
         (4096, 11008, 1), (4096, 11008, 4), (11008, 4096, 1), (11008, 4096, 4), (32768, 1536, 1), (4096, 4096, 1), (4096, 4096, 2), (2048, 2048, 1), (1024, 1024, 1)]
TR = os.environ.get("SWEEP_TR") == "1"
DTYPES = [np.float32, np.float16] if os.environ.get("SWEEP_F16") == "1" else [np.float32]
if len(sys.argv) > 1:
    CASES = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for dt in DTYPES:
    for R, C, n in CASES:
        vlen, olen = (R, C) if TR else (C, R)
        A = B.device_random(wg, gpu, (R, C), dt, 1)
        x = B.device_random(wg, gpu, (vlen, n), dt, 2)
        y = wg.TensorBuilder.matrix(olen, n, S.STORAGE | S.COPY_SRC).build(gpu.device(), dt)
        gemv = wg.Gemv.from_device(gpu.device())
        shapes = wg.ViewShapeBuffers()

        def run(k):
            enc = gpu.device().create_command_encoder()
            p = enc.compute_pass("gemv", None)
            for _ in range(k):
                gemv.dispatch_generic(gpu.device(), shapes, p, y.as_embedded_view(2), A.as_embedded_view(3), x.as_embedded_view(2), wg.GemvVariant.GemvTr if TR else wg.GemvVariant.Gemv)
            p.end()
            gpu.queue().submit([enc.finish()])
            gpu.sync()
        run(10)
        if R * C <= 8192 * 8192:  # numbers too: against f64 on the host
            a64 = A.slow_read(gpu).astype(np.float64).reshape(C, R).T
            x64 = x.slow_read(gpu).astype(np.float64).reshape(n, vlen).T
            ref = a64.T @ x64 if TR else a64 @ x64
            got = y.slow_read(gpu).astype(np.float64).reshape(n, olen).T
            err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
            assert err < (2e-3 if dt == np.float16 else 1e-5), (R, C, n, dt, err)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            run(200)
            best = min(best, (time.perf_counter() - t0) / 200)
        byts = np.dtype(dt).itemsize * (R * C + n * (R + C))
        print(f"{'gemv_tr' if TR else 'gemv'} {np.dtype(dt).name} {R}x{C}x{n}: {best*1e6:8.1f} us {byts/best/1e9:6.0f} GB/s", flush=True)
        del A, x, y
