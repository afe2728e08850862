#!/usr/bin/env python3
"""Vendor-library yardstick (NOT part of the product or the bench): torch.matmul (hipBLASLt / rocBLAS) on the same shapes and
the same U[-1,1) f16/f32 operands as bench.py, timed with the same sustained-run method. Used only to judge how far the
hand-written kernels are from what the vendor's tuned assembly reaches under the same power cap."""
import sys, time, torch

def run(n, dtype, layout, seconds=1.5):
    a = (torch.rand(n, n, device="cuda", dtype=torch.float32) * 2 - 1).to(dtype)
    b = (torch.rand(n, n, device="cuda", dtype=torch.float32) * 2 - 1).to(dtype)
    if layout == "nt": b = b.t().contiguous().t()
    if layout == "tn": a = a.t().contiguous().t()
    c = torch.empty(n, n, device="cuda", dtype=dtype)
    for _ in range(3): torch.matmul(a, b, out=c)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); torch.matmul(a, b, out=c); torch.cuda.synchronize(); one = time.perf_counter() - t0
    iters = max(5, int(seconds / one))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): torch.matmul(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * n ** 3 / ms / 1e9

for dtype, sizes in ((torch.float16, (4096, 8192, 16384)), (torch.bfloat16, (8192,)), (torch.float32, (4096,))):
    for n in sizes:
        for layout in ("nn", "nt", "tn"):
            print(f"{str(dtype):16s} {n:6d}^3 {layout}: {run(n, dtype, layout):8.1f} TFLOP/s", flush=True)
