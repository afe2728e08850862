#!/bin/bash
# round 3, GPU job e: K remainder + unaligned-view staging parity, K % 64 != 0 throughput
mkdir -p gpurun_out/r03e; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f16 or unaligned or errors or strided" > $O/pytest_f16.txt 2>&1
python - > $O/oddk.txt 2>&1 <<'PY'
import time, numpy as np, wgmath_amd as wg
from bench import device_random
gpu = wg.GpuInstance.new(0); dev, shapes = gpu.device(), wg.ViewShapeBuffers(); S = wg.BufferUsages
gemm = wg.Gemm.from_device(dev)
for (M, N, K) in [(8192, 8192, 8192), (8192, 8192, 8224), (8192, 8192, 8200), (8192, 8192, 8248), (4096, 4096, 4100 + 4), (4096, 4096, 4096 + 32), (8192, 8192, 512 + 32), (8192,8192,2048+8)]:
    for tr in (False, True):
        a = device_random(wg, gpu, (K, M) if tr else (M, K), np.float16, 1); b = device_random(wg, gpu, (K, N), np.float16, 2)
        c = wg.TensorBuilder.matrix(M, N, S.STORAGE).build(dev, np.float16)
        enc = dev.create_command_encoder(); p = enc.compute_pass("t", None)
        v = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
        for _ in range(10): gemm.dispatch_generic(dev, shapes, p, c, a, b, v)
        gpu.sync(); n = 100; t0 = time.perf_counter()
        for _ in range(n): gemm.dispatch_generic(dev, shapes, p, c, a, b, v)
        gpu.sync(); dt = (time.perf_counter() - t0) / n
        print(f"{M}x{N}x{K} tr={tr}: {dt*1e6:9.1f} us  {2.0*M*N*K/dt/1e12:8.1f} TFLOP/s", flush=True)
PY
