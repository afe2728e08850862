import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import wgmath_amd as wg
from bench import device_random
gpu = wg.GpuInstance.new(0); dev, shapes = gpu.device(), wg.ViewShapeBuffers(); S = wg.BufferUsages
gemm = wg.Gemm.from_device(dev)
for (M, N, K) in [(8192, 8192, 8192), (8192, 8192, 8224), (8192, 8192, 8200), (8192, 8192, 8160), (4096, 4096, 4100), (8192, 8192, 1056)]:
    a = device_random(wg, gpu, (M, K), np.float16, 1); b = device_random(wg, gpu, (K, N), np.float16, 2)
    c = wg.TensorBuilder.matrix(M, N, S.STORAGE).build(dev, np.float16)
    enc = dev.create_command_encoder(); p = enc.compute_pass("t", None)
    for _ in range(3): gemm.dispatch(dev, shapes, p, c, a, b)
    gpu.sync(); t0 = time.perf_counter(); n = 20
    for _ in range(n): gemm.dispatch(dev, shapes, p, c, a, b)
    gpu.sync(); dt = (time.perf_counter() - t0) / n
    print(f"{M}x{N}x{K}: {dt*1e6:9.1f} us  {2.0*M*N*K/dt/1e12:8.1f} TFLOP/s", flush=True)
    del a, b, c
