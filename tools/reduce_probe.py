import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, wgmath_amd as wg
from bench import device_random
gpu = wg.GpuInstance.new(0); dev = gpu.device(); shapes = wg.ViewShapeBuffers(); S = wg.BufferUsages
enc = dev.create_command_encoder(); p = enc.compute_pass("x", None)
x = device_random(wg, gpu, (1 << 26,), np.float32, 1)
res = wg.TensorBuilder.scalar(S.STORAGE | S.COPY_SRC).build(dev, np.float32)
red = wg.Reduce.new(dev, wg.ReduceOp.Sum)
for k in (10, 14, 16, 18, 20, 22, 24, 26):
    n = 1 << k
    v = x.rows(0, n)
    for _ in range(3): red.dispatch(dev, shapes, p, v, res)
    reps = max(3, min(200, (1 << 22) // n + 3))
    ts = wg.GpuTimestamps.new(dev, 2); ts.write(dev)
    for _ in range(reps): red.dispatch(dev, shapes, p, v, res)
    ts.write(dev); t = ts.wait_for_results_ms(); dt = (t[1] - t[0]) / reps * 1e-3
    print(f"n=2^{k}: {dt*1e6:10.1f} us  {4*n/dt/1e9:8.2f} GB/s")
