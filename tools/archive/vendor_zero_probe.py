#!/usr/bin/env python3
"""Vendor GEMM (torch.matmul -> hipBLASLt) on ZERO and on random f16 operands, 8192^3, all three layouts: with zeros the chip is off its
power cap, the clock pins at its maximum and the time measures cycles -- the yardstick for tools/ab.sh runs with WG_BENCH_VALUES=zero."""
import time, torch
n = 8192
for fill in ("zero", "random"):
    for layout in ("nn", "nt", "tn"):
        a = torch.zeros(n, n, device="cuda", dtype=torch.float16) if fill == "zero" else (torch.rand(n, n, device="cuda") * 2 - 1).half()
        b = torch.zeros(n, n, device="cuda", dtype=torch.float16) if fill == "zero" else (torch.rand(n, n, device="cuda") * 2 - 1).half()
        if layout == "nt": b = b.t().contiguous().t()
        if layout == "tn": a = a.t().contiguous().t()
        c = torch.empty(n, n, device="cuda", dtype=torch.float16)
        for _ in range(50): torch.matmul(a, b, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(1000): torch.matmul(a, b, out=c)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 1000
        print(f"vendor {fill:6s} {layout}: {ms * 1e3:7.1f} us  {2.0 * n ** 3 / ms / 1e9:7.1f} TFLOP/s", flush=True)
