#!/usr/bin/env python3
"""Debug aid: forced balance units vs the static launch on one shape; which tiles differ, and what do they look like (missing prefix? stale?)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wgmath_amd as wg
from wgmath_amd import _lib
M, K, N = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 512, 4096)
gpu = wg.GpuInstance.new(0); dev, shapes = gpu.device(), wg.ViewShapeBuffers(); S = wg.BufferUsages
U = S.STORAGE | S.COPY_SRC | S.COPY_DST
rng = np.random.default_rng(5)
gemm = wg.Gemm.from_device(dev)
def up(shape, arr): return wg.TensorBuilder.tensor(shape, U).build_init(dev, arr)
def run(a_t, b_t):
    out = up((M, N, 1), np.full(M * N, np.nan, np.float16))
    enc = dev.create_command_encoder(); p = enc.compute_pass("t", None)
    gemm.dispatch(dev, shapes, p, out, a_t, b_t); p.end(); gpu.queue().submit([enc.finish()])
    return out.read(dev).reshape(N, M).T.copy()
gpu.set_tuning("f16_tile", 256); gpu.set_tuning("f16_sched", 0)
stages = K // 64
tiles = (M // 256) * (N // 256)
cap = 2 * tiles + 1024
units = (ctypes.c_uint32 * (5 * cap))(); n = ctypes.c_uint32(); nwg = ctypes.c_uint32()
_lib.check(_lib.lib.wg_debug_f16_balance_plan((ctypes.c_double * 8)(*([1.0] * 8)), tiles, stages, 1, units, cap, ctypes.byref(n), ctypes.byref(nwg)))
u = np.array(units[:5 * n.value]).reshape(-1, 5)
suf = {int(r[0]): r for r in u if r[1] == 2}
print("plan:", len(u), "units,", (u[:, 1] == 1).sum(), "prefix; suffix tiles:", sorted(suf)[:40])
for it in range(3):
    a = (rng.random(M * K, dtype=np.float32) * 2 - 1).astype(np.float16); b = (rng.random(K * N, dtype=np.float32) * 2 - 1).astype(np.float16)
    ta, tb = up((M, K, 1), a), up((K, N, 1), b)
    gpu.set_tuning("f16_balance", 0); ref = run(ta, tb)
    gpu.set_tuning("f16_balance", 1); got = run(ta, tb)
    A = a.reshape(K, M).T.astype(np.float64); B = b.reshape(N, K).T.astype(np.float64)
    bad = []
    tm_n = M // 256
    for tn in range(N // 256):
        for tm in range(tm_n):
            if not np.array_equal(ref[tm*256:(tm+1)*256, tn*256:(tn+1)*256].view(np.uint16), got[tm*256:(tm+1)*256, tn*256:(tn+1)*256].view(np.uint16)):
                bad.append((tm, tn))
    print(f"iteration {it}: {len(bad)} differing tiles (by tm, tn): {bad[:16]}  info {gpu.f16_balance_info()}")
    for (tm, tn) in bad[:3]:
        g = got[tm*256:(tm+1)*256, tn*256:(tn+1)*256].astype(np.float64); r = ref[tm*256:(tm+1)*256, tn*256:(tn+1)*256].astype(np.float64)
        At, Bt = A[tm*256:(tm+1)*256], B[:, tn*256:(tn+1)*256]
        for kb in range(0, stages + 1):
            sfx = At[:, kb*64:] @ Bt[kb*64:]
            if np.abs(g - sfx).max() < 0.05 and kb > 0: print(f"   tile {(tm,tn)} looks like the SUFFIX ALONE from stage {kb} (max |got - suffix| {np.abs(g - sfx).max():.3g}; max |got - ref| {np.abs(g - r).max():.3g})")
        full = A @ B
        best = None
        for tn2 in range(N // 256):
            for tm2 in range(M // 256):
                e = np.abs(g - full[tm2*256:(tm2+1)*256, tn2*256:(tn2+1)*256]).max()
                if best is None or e < best[0]: best = (e, tm2, tn2)
        print(f"   tile {(tm,tn)}: closest full tile of the true product: {best}")
        # which rows / cols of A, B does it correlate with?
        ca = [(np.corrcoef(g.ravel(), full[tm2*256:(tm2+1)*256, tn*256:(tn+1)*256].ravel())[0, 1], tm2) for tm2 in range(M // 256)]
        cb = [(np.corrcoef(g.ravel(), full[tm*256:(tm+1)*256, tn2*256:(tn2+1)*256].ravel())[0, 1], tn2) for tn2 in range(N // 256)]
        print("      corr with same column-tile, other tm:", max(ca), " same row-tile, other tn:", max(cb))
        for (tm2, tn2) in [(tm, tn + 6), (tm, tn)]:
            At2, Bt2 = A[tm2*256:(tm2+1)*256], B[:, tn2*256:(tn2+1)*256]
            for kb in range(1, stages):
                for nm, prod in (("prefix", At2[:, :kb*64] @ Bt2[:kb*64]), ("suffix", At2[:, kb*64:] @ Bt2[kb*64:])):
                    if np.abs(g - prod).max() < 0.1: print(f"   tile {(tm,tn)} == {nm} of tile {(tm2,tn2)} at stage {kb}")
        print(f"   tile {(tm,tn)}: nan count {np.isnan(g).sum()}, max|got-ref| {np.nanmax(np.abs(g-r)):.4g}, rows differing {np.unique(np.where(g!=r)[0]).size}, cols differing {np.unique(np.where(g!=r)[1]).size}")
