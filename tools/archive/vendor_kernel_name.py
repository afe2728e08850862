#!/usr/bin/env python3
"""Which hipBLASLt kernel the vendor library picks for f16 8192^3 (and 2048^3 / f32 2048^3) on its best layout and on this repo's layout.
Run under `rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/vendor_kernel_name.py`: the kernel names land in the trace; this
script prints the rates it measures itself (torch events) so that name and rate come from the same run."""
import time

import torch


def run(n, layout, dtype=torch.float16, seconds=0.5):
    a = (torch.rand(n, n, device="cuda", dtype=torch.float32) * 2 - 1).to(dtype)
    b = (torch.rand(n, n, device="cuda", dtype=torch.float32) * 2 - 1).to(dtype)
    # torch is row-major: C = A @ B with all three row-major is, read column-major, C^T = B^T A^T -- the vendor's "NN"; a B stored transposed
    # (b.t().contiguous().t()) makes the k index contiguous in both operands -- the layout its best kernels are written for ("TN" in BLAS terms,
    # GemmTr in this repo's)
    if layout == "tn":
        b = b.t().contiguous().t()
    c = torch.empty(n, n, device="cuda", dtype=dtype)
    for _ in range(3):
        torch.matmul(a, b, out=c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.matmul(a, b, out=c)
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    iters = max(5, int(seconds / one))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n ** 3 / (e0.elapsed_time(e1) / iters) / 1e9


import sys  # noqa: E402

for n, dt in () if len(sys.argv) > 1 else ((8192, torch.float16), (2048, torch.float16), (4096, torch.float16), (2048, torch.float32), (1024, torch.float32)):
    for layout in ("nn", "tn"):
        print(f"vendor {str(dt).split('.')[-1]} {n}^3 {layout}: {run(n, layout, dt):8.1f} TFLOP/s", flush=True)


def run_rect(M, N, K, seconds=0.3):
    """GemmTr's layout (both operands k-contiguous), rectangular: column-major C (M x N) = row-major (N x K) @ (M x K)^T."""
    x = (torch.rand(N, K, device="cuda") * 2 - 1).half()
    y = (torch.rand(M, K, device="cuda") * 2 - 1).half().t()
    c = torch.empty(N, M, device="cuda", dtype=torch.float16)
    for _ in range(5):
        torch.matmul(x, y, out=c)
    torch.cuda.synchronize()
    iters = max(20, int(seconds / (2.0 * M * N * K / 1.0e15 + 5e-6)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(x, y, out=c)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for s in sys.argv[1:]:  # extra shapes MxNxK on the k-contiguous layout
    M, N, K = (int(v) for v in s.split("x"))
    print(f"vendor f16 tn {M}x{N}x{K}: {run_rect(M, N, K):8.1f} us", flush=True)
