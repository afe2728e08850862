#!/usr/bin/env python3
"""Timing experiment (trace build of the library: make EXTRA=-DWG_F16_TRACE=1): per-workgroup timestamps of the f16 GEMM.
usage: WGEBRA_HIP_LIB=.../libwgebra_hip_trace2.so python tools/f16_trace.py M K N [zero] [tn]
-DWG_F16_TRACE=2 builds add, per wave, the split of every end-of-half-step into counted-DMA wait / barrier wait (shader cycles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wgmath_amd as wg
from bench import device_random

M, K, N = (int(x) for x in sys.argv[1:4])
gpu = wg.GpuInstance.new(0)
dev, shapes = gpu.device(), wg.ViewShapeBuffers()
S = wg.BufferUsages
ZERO, TN = "zero" in sys.argv[4:], "tn" in sys.argv[4:]
if ZERO:
    a = wg.TensorBuilder.matrix(*((K, M) if TN else (M, K)), S.STORAGE).build(dev, np.float16)
    b = wg.TensorBuilder.matrix(K, N, S.STORAGE).build(dev, np.float16)
else:
    a = device_random(wg, gpu, (K, M) if TN else (M, K), np.float16, 1)
    b = device_random(wg, gpu, (K, N), np.float16, 2)
c = wg.TensorBuilder.matrix(M, N, S.STORAGE).build(dev, np.float16)
gemm = wg.Gemm.from_device(dev)
variant = wg.GemmVariant.GemmTr if TN else wg.GemmVariant.Gemm
enc = dev.create_command_encoder(); p = enc.compute_pass("t", None)
for _ in range(int(os.environ.get("WG_TRACE_REPS", "6"))): gemm.dispatch_generic(dev, shapes, p, c, a, b, variant)
gpu.sync()
raw = np.fromfile("/tmp/wg_f16_trace.bin", dtype=np.uint64)
ntile = (M // 256) * (N // 256)
fine = None
if len(raw) == ntile * 24:
    fine = raw[ntile * 8:].view(np.uint32).reshape(ntile, 4, 8).astype(np.float64)
    raw = raw[:ntile * 8]
t = raw.reshape(-1, 8)
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
print(f"  distinct (xcc, se, sh, cu) keys: {len(np.unique(key))}; xcc values {np.unique(xcc)}")
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
loop_us = ts[:, 2] - ts[:, 1]
print("  per XCD: tiles, mean / min / max loop us, last tile end us:")
for x in np.unique(xcc):
    m = xcc == x
    print(f"    xcd {x}: {m.sum():5d} tiles  {loop_us[m].mean():8.2f} {loop_us[m].min():8.2f} {loop_us[m].max():8.2f}   end {ts[m, 4].max():10.1f}")
ends = np.array([ts[key == k, 4].max() for k in np.unique(key)])
print(f"  last tile end per CU: min {ends.min():.1f}  mean {ends.mean():.1f}  max {ends.max():.1f} us  -> idle at the end, mean over CUs: {ends.max() - ends.mean():.1f} us ({100 * (ends.max() - ends.mean()) / ends.max():.2f} %)")
starts = np.sort(ts[:, 0])
print("  start times of the first 8 and of workgroups 256..263:", np.round(starts[:8], 2), np.round(starts[256:264], 2) if len(starts) > 264 else "")

if fine is not None and fine[:, :, 0].sum() > 0:
    loop, vm, bar, probe, mvm, mbar, nadv = (fine[:, :, i] for i in range(7))
    n = nadv.mean()
    hs = K // 32                      # half-steps per tile; each 64 MFMAs per wave = 1024 cycles at 16 cycles / MFMA
    work = loop - vm - bar - probe    # cycles spent issuing the half-steps themselves
    print(f"fine trace ({'zero' if ZERO else 'random'} operands, {'TN' if TN else 'NN'}): {hs} half-steps per tile, {n:.0f} instrumented ends; shader cycles per wave, mean over {ntile} tiles x 4 waves")
    print(f"  loop span                {loop.mean():10.0f}   = {loop.mean()/hs:7.1f} per half-step (ideal 1024)")
    print(f"  half-step bodies         {work.mean():10.0f}   = {work.mean()/hs:7.1f} per half-step  ({100*work.mean()/loop.mean():.1f} % of the loop)")
    print(f"  counted-DMA wait (vmcnt) {vm.mean():10.0f}   = {vm.mean()/n:7.1f} per end   ({100*vm.mean()/loop.mean():.1f} %)   max single {mvm.max():.0f}, mean of per-wave max {mvm.mean():.0f}")
    print(f"  barrier wait             {bar.mean():10.0f}   = {bar.mean()/n:7.1f} per end   ({100*bar.mean()/loop.mean():.1f} %)   max single {mbar.max():.0f}, mean of per-wave max {mbar.mean():.0f}")
    print(f"  probe round trips        {probe.mean():10.0f}   = {probe.mean()/n:7.1f} per end   ({100*probe.mean()/loop.mean():.1f} %; not in the shipped kernel)")
    for w in range(4):
        print(f"    wave {w}: body {work[:, w].mean()/hs:7.1f}  vm {vm[:, w].mean()/n:6.1f}  barrier {bar[:, w].mean()/n:6.1f}")
    d = ts[:, 2] - ts[:, 1]
    print(f"  loop span in us (100 MHz clock) {d.mean():.2f} -> shader clock {loop.mean()/d.mean()/1000:.3f} GHz")
