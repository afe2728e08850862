#!/usr/bin/env python3
"""Per-launch efficiency of short GEMM launches (a few rounds of tiles) on a full and on a CU-masked stream."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wgmath_amd as wg
from bench import device_random
K = 32768
for cus, M, N in ((256, 8192, 8192), (256, 8192, 4096), (224, 8192, 7168), (224, 8192, 3584), (240, 8192, 3840), (240, 8192, 7680)):
    g = wg.GpuInstance.new(0, cu_count=cus) if cus != 256 else wg.GpuInstance.new(0)
    d = g.device(); S = wg.BufferUsages
    a = device_random(wg, g, (M, K), np.float16, 1); b = device_random(wg, g, (K, N), np.float16, 2)
    c = wg.TensorBuilder.matrix(M, N, S.STORAGE).build(d, np.float16)
    gemm, shapes = wg.Gemm.from_device(d), wg.ViewShapeBuffers()
    p = d.create_command_encoder().compute_pass("g", None)
    for _ in range(3): gemm.dispatch(d, shapes, p, c, a, b)
    g.sync(); t0 = time.perf_counter()
    for _ in range(20): gemm.dispatch(d, shapes, p, c, a, b)
    g.sync(); dt = (time.perf_counter() - t0) / 20
    tiles = (M // 256) * (N // 256)
    print(f"{cus} CUs, {tiles} tiles/launch ({tiles / cus:.2f} rounds): {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.0f} TFLOP/s")
    del a, b, c, g
