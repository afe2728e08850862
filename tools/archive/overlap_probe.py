#!/usr/bin/env python3
"""How does the chip interleave a second queue's small kernel with a resident f16 GEMM grid (every GEMM workgroup needs a whole
CU)? Stand-in for RCCL's copy kernels next to the M-sharded panel GEMMs of bench.py --gpus N.

Context 1 (stream 1): the panel GEMM of the 4-GPU configuration (8192 x 4096 x 32768 = 512 tiles, 2 rounds).
Context 2 (stream 2): `blocks` workgroups of 256 threads that spin for `usec` us (wg_debug_spin), enqueued right after the GEMM.
Reports: when the spinners started relative to the GEMM's start, and the GEMM's duration with / without them."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wgmath_amd as wg
from wgmath_amd._lib import lib, check
from bench import device_random

M, N, K = 8192, 4096, 32768
CUS = int(sys.argv[1]) if len(sys.argv) > 1 else 0        # > 0: the GEMM's stream is CU-masked to that many CUs
g1 = wg.GpuInstance.new(0, cu_count=CUS) if CUS else wg.GpuInstance.new(0)
g2 = wg.GpuInstance.new(0)                                # second context = second stream on the same device
print("GEMM stream CUs:", CUS or 256)
d1 = g1.device()
S = wg.BufferUsages
a = device_random(wg, g1, (M, K), np.float16, 1)
b = device_random(wg, g1, (K, N), np.float16, 2)
c = wg.TensorBuilder.matrix(M, N, S.STORAGE).build(d1, np.float16)
gemm, shapes = wg.Gemm.from_device(d1), wg.ViewShapeBuffers()
p1 = d1.create_command_encoder().compute_pass("gemm", None)

def gemm_ms(spin_blocks, spin_usec, reps=5):
    ticks = wg.TensorBuilder.vector(2 * (spin_blocks + 2), S.STORAGE | S.COPY_SRC).build(g2.device(), np.float32) if spin_blocks else None
    t0buf = wg.TensorBuilder.vector(4, S.STORAGE | S.COPY_SRC).build(d1, np.float32)
    out = []
    for _ in range(reps):
        g1.sync(); g2.sync()
        ts = wg.GpuTimestamps.new(d1, 2)
        ts.write(d1)
        check(lib.wg_debug_spin(g1._ctx.handle, 1, 0, t0buf._h))   # tick just before the GEMM on stream 1
        gemm.dispatch(d1, shapes, p1, c, a, b)
        if spin_blocks:
            check(lib.wg_debug_spin(g2._ctx.handle, spin_blocks, spin_usec, ticks._h))
        ts.write(d1)
        g1.sync(); g2.sync()
        t = ts.wait_for_results_ms()
        res = {"gemm_ms": t[1] - t[0]}
        if spin_blocks:
            tk = ticks.read(g2.device()).view(np.uint64)
            t0 = t0buf.read(d1).view(np.uint64)[0]
            st = (tk[:spin_blocks].astype(np.int64) - np.int64(t0)) / 100.0  # us after the GEMM's start
            res.update(spin_first_us=float(st.min()), spin_median_us=float(np.median(st)), spin_last_us=float(st.max()))
        out.append(res)
    return out[-1]

print("GEMM alone:", gemm_ms(0, 0))
for blocks, usec in ((16, 1500), (32, 1500), (64, 1500), (32, 300)):
    print(f"GEMM + {blocks} spinning workgroups x {usec} us on a second stream:", gemm_ms(blocks, usec))
