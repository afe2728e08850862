import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, wgmath_amd as wg
from bench import device_random
gpu = wg.GpuInstance.new(0); dev = gpu.device(); shapes = wg.ViewShapeBuffers()
enc = dev.create_command_encoder(); p = enc.compute_pass("x", None)
n = 1 << 28
a = device_random(wg, gpu, (n,), np.float32, 1); b = device_random(wg, gpu, (n,), np.float32, 2)
def timeit(fn, reps=200):
    for _ in range(10): fn()
    ts = wg.GpuTimestamps.new(dev, 2); ts.write(dev)
    for _ in range(reps): fn()
    ts.write(dev); t = ts.wait_for_results_ms(); return (t[1]-t[0])/reps*1e-3
for op, nbytes in ((wg.OpAssignVariant.Add, 12), (wg.OpAssignVariant.Mul, 12), (wg.OpAssignVariant.Div, 12), (wg.OpAssignVariant.Copy, 8)):
    o = wg.OpAssign.new(dev, op)
    dt = timeit(lambda: o.dispatch(dev, shapes, p, a, b))
    print(f"{op.name:5s} {nbytes*n/dt/1e9:8.0f} GB/s  {dt*1e3:.3f} ms")
ax = wg.Axpy.from_device(dev)
dt = timeit(lambda: ax.dispatch(dev, shapes, p, 0.5, a, b)); print(f"axpy  {12*n/dt/1e9:8.0f} GB/s")
# smaller sizes
for k in (20, 24, 26):
    m = 1 << k
    o = wg.OpAssign.new(dev, wg.OpAssignVariant.Add)
    dt = timeit(lambda: o.dispatch(dev, shapes, p, a.rows(0, m), b.rows(0, m)), reps=500)
    print(f"Add n=2^{k}: {12*m/dt/1e9:8.0f} GB/s {dt*1e6:.1f} us")
