#!/usr/bin/env python3
"""Vendor-library yardsticks for the HBM-bound operators (NOT part of the product or the bench): torch.mv / torch.sum / torch add_ / copy_
(rocBLAS gemv, PyTorch's reduction and elementwise kernels) on the shapes of bench.py, same sustained-run timing."""
import time, torch

def timeit(fn, seconds=1.0):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); one = max(time.perf_counter() - t0, 1e-6)
    n = max(10, int(seconds / one))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

R, C = 4096, 65536
m = torch.rand(C, R, device="cuda").t()              # column-major R x C (what the library's Gemv sees)
v, vt = torch.rand(C, device="cuda"), torch.rand(R, device="cuda")
out, outt = torch.empty(R, device="cuda"), torch.empty(C, device="cuda")
b = 4.0 * (R * C + R + C)
print(f"gemv   N 4096x65536 (col-major): {b / timeit(lambda: torch.mv(m, v, out=out)) / 1e9:7.0f} GB/s")
print(f"gemv   T 65536x4096 (same data) : {b / timeit(lambda: torch.mv(m.t(), vt, out=outt)) / 1e9:7.0f} GB/s")
x = torch.rand(4096, 65536, device="cuda")
res = torch.empty(4096, device="cuda")
print(f"reduce 4096 sums of 65536      : {4.0 * x.numel() / timeit(lambda: torch.sum(x, dim=1, out=res)) / 1e9:7.0f} GB/s")
a1, b1 = torch.rand(1 << 28, device="cuda"), torch.rand(1 << 28, device="cuda")
print(f"a += b, 2^28 f32               : {12.0 * a1.numel() / timeit(lambda: a1.add_(b1)) / 1e9:7.0f} GB/s")
print(f"a = b (copy), 2^28 f32         : {8.0 * a1.numel() / timeit(lambda: a1.copy_(b1)) / 1e9:7.0f} GB/s")
x1 = torch.rand(1 << 26, device="cuda")
print(f"sum of one 2^26 vector         : {4.0 * x1.numel() / timeit(lambda: torch.sum(x1)) / 1e9:7.0f} GB/s")
