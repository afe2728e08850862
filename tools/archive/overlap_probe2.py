#!/usr/bin/env python3
"""Back-to-back GEMM launches on stream 1 (the panel pipeline) and ONE copy-kernel stand-in on stream 2 enqueued right after the first
GEMM: when do the stand-in's workgroups start -- at the first launch boundary, or only when stream 1 runs dry?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wgmath_amd as wg
from wgmath_amd._lib import lib, check
from bench import device_random
CUS = int(sys.argv[1]) if len(sys.argv) > 1 else 0
NG = 6
M, K = 8192, 32768
N = 4096 if not CUS else (CUS * 2 // 32) * 256   # two rounds per launch
g1 = wg.GpuInstance.new(0, cu_count=CUS) if CUS else wg.GpuInstance.new(0)
g2 = wg.GpuInstance.new(0)
d1 = g1.device(); S = wg.BufferUsages
a = device_random(wg, g1, (M, K), np.float16, 1); b = device_random(wg, g1, (K, N), np.float16, 2)
c = wg.TensorBuilder.matrix(M, N, S.STORAGE).build(d1, np.float16)
gemm, shapes = wg.Gemm.from_device(d1), wg.ViewShapeBuffers()
p1 = d1.create_command_encoder().compute_pass("gemm", None)
ticks = wg.TensorBuilder.vector(2 * 40, S.STORAGE | S.COPY_SRC).build(g2.device(), np.float32)
t0buf = wg.TensorBuilder.vector(4, S.STORAGE | S.COPY_SRC).build(d1, np.float32)
for rep in range(3):
    g1.sync(); g2.sync()
    ts = wg.GpuTimestamps.new(d1, 2); ts.write(d1)
    check(lib.wg_debug_spin(g1._ctx.handle, 1, 0, t0buf._h))
    gemm.dispatch(d1, shapes, p1, c, a, b)
    check(lib.wg_debug_spin(g2._ctx.handle, 32, 1000, ticks._h))      # "all-gather of panel 0"
    for _ in range(NG - 1): gemm.dispatch(d1, shapes, p1, c, a, b)
    ts.write(d1); g1.sync(); g2.sync()
    t = ts.wait_for_results_ms()
    tk = ticks.read(g2.device()).view(np.uint64); t0 = t0buf.read(d1).view(np.uint64)[0]
    st = (tk[:32].astype(np.int64) - np.int64(t0)) / 100.0
print(f"compute stream CUs {CUS or 256}: {NG} GEMM launches of {(M//256)*(N//256)} tiles took {t[1]-t[0]:.2f} ms ({(t[1]-t[0])/NG:.2f} each); "
      f"stand-in workgroups started {st.min():.0f} / {np.median(st):.0f} / {st.max():.0f} us (first / median / last) after the first GEMM began")
