"""Batched Reduce, one-vector Reduce (reference order and the two-pass `fast` form) and OpAssign over a sweep of sizes, f32 and f16: GPU time per
dispatch from back-to-back eager launches, GB/s of algorithmic bytes. Usage (GPU box): python tools/hbm_ops_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench as B  # noqa: E402
import wgmath_amd as wg  # noqa: E402

gpu = wg.GpuInstance.new(0)
dev = gpu.device()
S = wg.BufferUsages
shapes = wg.ViewShapeBuffers()


def timed(fn, reps=200):
    def run(n):
        enc = dev.create_command_encoder()
        p = enc.compute_pass("x", None)
        for _ in range(n):
            fn(p)
        p.end()
        gpu.queue().submit([enc.finish()])
        gpu.sync()
    run(10)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        run(reps)
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


for dt in (np.float32, np.float16):
    es = np.dtype(dt).itemsize
    name = np.dtype(dt).name
    for n, cols in [(65536, 4096), (16384, 4096), (4096, 4096), (1024, 4096), (345, 4096), (128, 65536), (1024, 65536), (4096, 512), (65536, 64), (1048576, 16), (1000, 1000), (100003, 37)]:
        X = B.device_random(wg, gpu, (n, cols), dt, 3)
        R = wg.TensorBuilder.vector(cols, S.STORAGE | S.COPY_SRC).build(dev, dt)
        red = wg.Reduce.new(dev, wg.ReduceOp.Sum)
        t = timed(lambda p: red.dispatch_batched(dev, shapes, p, X, R))
        print(f"reduce_batched {name} {n}x{cols}: {t*1e6:8.1f} us {es*(n*cols+cols)/t/1e9:6.0f} GB/s", flush=True)
        del X, R
    for n in [1 << 16, 1 << 20, 1 << 24, 1 << 26, 1 << 28]:
        x = B.device_random(wg, gpu, (n,), dt, 4)
        r = wg.TensorBuilder.scalar(S.STORAGE | S.COPY_SRC).build(dev, dt)
        for op in (wg.ReduceOp.Sum, wg.ReduceOp.Max):
            red = wg.Reduce.new(dev, op)
            t = timed(lambda p: red.dispatch(dev, shapes, p, x, r), reps=50 if n >= 1 << 24 else 200)
            t2 = timed(lambda p: red.dispatch_fast(dev, shapes, p, x, r))
            print(f"reduce {op.name} {name} n={n}: reference order {t*1e6:9.1f} us {es*n/t/1e9:6.0f} GB/s | fast {t2*1e6:8.1f} us {es*n/t2/1e9:6.0f} GB/s", flush=True)
        del x, r
    for n in [1757, 1 << 16, 1 << 20, 1 << 22, 1 << 24, 1 << 26, (1 << 28), 100000003]:
        a = B.device_random(wg, gpu, (n,), dt, 5)
        b = B.device_random(wg, gpu, (n,), dt, 6)
        for op in (wg.OpAssignVariant.Add, wg.OpAssignVariant.Copy):
            oa = wg.OpAssign.new(dev, op)
            t = timed(lambda p: oa.dispatch(dev, shapes, p, a, b))
            byts = es * n * (3 if op != wg.OpAssignVariant.Copy else 2)
            print(f"op_assign {op.name} {name} n={n}: {t*1e6:8.1f} us {byts/t/1e9:6.0f} GB/s", flush=True)
        del a, b
