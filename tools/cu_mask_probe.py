#!/usr/bin/env python3
"""Is a CU-masked stream (wg_ctx_create_with_cu_count) balanced over the XCDs? 3840 tiles = 15 rounds of 256 = 16 rounds of 240."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wgmath_amd as wg
from bench import device_random
M, N, K = 8192, 30720, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for cus in (256, 248, 240, 224, 192):
    g = wg.GpuInstance.new(0, cu_count=cus) if cus != 256 else wg.GpuInstance.new(0)
    d = g.device(); S = wg.BufferUsages
    a = device_random(wg, g, (M, K), np.float16, 1); b = device_random(wg, g, (K, N), np.float16, 2)
    c = wg.TensorBuilder.matrix(M, N, S.STORAGE).build(d, np.float16)
    gemm, shapes = wg.Gemm.from_device(d), wg.ViewShapeBuffers()
    p = d.create_command_encoder().compute_pass("g", None)
    for _ in range(3): gemm.dispatch(d, shapes, p, c, a, b)
    g.sync(); t0 = time.perf_counter()
    for _ in range(10): gemm.dispatch(d, shapes, p, c, a, b)
    g.sync(); dt = (time.perf_counter() - t0) / 10
    print(f"{cus} CUs: {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.0f} TFLOP/s  ({2.0*M*N*K/dt/1e12/cus*256:.0f} per 256 CUs)")
    del a, b, c, g
