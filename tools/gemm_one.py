"""One Gemm shape, this library only (no torch: safe under rocprofv3). Usage: python tools/gemm_one.py f32|f16 n|t MxNxK [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import wgmath_amd as wg  # noqa: E402
from bench import device_random  # noqa: E402

dt = np.float16 if sys.argv[1] == "f16" else np.float32
tr = sys.argv[2] == "t"
gpu = wg.GpuInstance.new(0)
dev, shapes = gpu.device(), wg.ViewShapeBuffers()
gemm = wg.Gemm.from_device(dev)
for spec in sys.argv[3:]:
    if "x" not in spec:
        continue
    M, N, K = (int(x) for x in spec.split("x"))
    a = device_random(wg, gpu, (K, M) if tr else (M, K), dt, 1)
    b = device_random(wg, gpu, (K, N), dt, 2)
    c = wg.TensorBuilder.matrix(M, N, wg.BufferUsages.STORAGE).build(dev, dt)
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
    n = 20
    t0 = time.perf_counter()
    run(n)
    t = (time.perf_counter() - t0) / n
    print(f"{sys.argv[1]} {'gemm_tr' if tr else 'gemm'} {M}x{N}x{K}: {t*1e6:9.1f} us {2.0*M*N*K/t/1e12:7.1f} TF", flush=True)
