#!/usr/bin/env python3
"""Timing experiment (trace build of the library: make EXTRA=-DWG_F16_TRACE=1): per-workgroup timestamps of the f16 GEMM.
usage: WGEBRA_HIP_LIB=.../libwg_trace.so python tools/f16_trace.py M K N"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wgmath_amd as wg
from bench import device_random

M, K, N = (int(x) for x in sys.argv[1:4])
gpu = wg.GpuInstance.new(0)
dev, shapes = gpu.device(), wg.ViewShapeBuffers()
S = wg.BufferUsages
a = device_random(wg, gpu, (M, K), np.float16, 1)
b = device_random(wg, gpu, (K, N), np.float16, 2)
c = wg.TensorBuilder.matrix(M, N, S.STORAGE).build(dev, np.float16)
gemm = wg.Gemm.from_device(dev)
enc = dev.create_command_encoder(); p = enc.compute_pass("t", None)
for _ in range(6): gemm.dispatch(dev, shapes, p, c, a, b)
gpu.sync()
t = np.fromfile("/tmp/wg_f16_trace.bin", dtype=np.uint64).reshape(-1, 8)
ts = t[:, :5].astype(np.float64) * 0.01  # us (100 MHz)
t0 = ts[:, 0].min()
ts -= t0
hw = t[:, 5]
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7  # gfx9 HW_ID: CU_ID[11:8], SH_ID[12], SE_ID[15:13]
xcc = t[:, 6] & 0xF
key = xcc * 1000 + se * 100 + sh * 16 + cu
print(f"{len(t)} workgroups; kernel span {ts[:,4].max():.1f} us")
d = ts[:, 1:] - ts[:, :-1]
names = ["prologue (start -> first MFMA)", "main loop", "epilogue issue (cvt + stores)", "store drain (vmcnt(0))"]
for i, n in enumerate(names):
    print(f"  {n:34s} mean {d[:,i].mean():8.2f} us   min {d[:,i].min():8.2f}   max {d[:,i].max():8.2f}")
print(f"  workgroup lifetime                 mean {(ts[:,4]-ts[:,0]).mean():8.2f} us")
# gap between consecutive workgroups on the same CU
gaps = []
for k in np.unique(key):
    idx = np.where(key == k)[0]
    idx = idx[np.argsort(ts[idx, 0])]
    for i, j in zip(idx[:-1], idx[1:]):
        gaps.append(ts[j, 0] - ts[i, 4])
if gaps:
    gaps = np.array(gaps)
    print(f"  gap: end of a workgroup -> start of the next on the same CU: mean {gaps.mean():.2f} us  min {gaps.min():.2f}  max {gaps.max():.2f}  (n={len(gaps)}, {len(np.unique(key))} CUs)")
starts = np.sort(ts[:, 0])
print("  start times of the first 8 and of workgroups 256..263:", np.round(starts[:8], 2), np.round(starts[256:264], 2) if len(starts) > 264 else "")
