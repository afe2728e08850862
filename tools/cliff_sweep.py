"""Looks for performance cliffs next to the tuned shapes: a base Gemm / Gemv shape against the same shape with one dimension (or one view's offset / leading
dimension) nudged off the alignments the fast kernels want (multiples of 4 elements = the reference's vec4 contract, of 8 = 16 bytes of f16, of the tile sizes).
Prints us per dispatch and the ratio to the base shape; anything far above 1 is a path that copies or falls to a slow kernel.
Usage (GPU box): [CLIFF_ONLY=gemv|gemm|vec] python tools/cliff_sweep.py [f32|f16 ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import wgmath_amd as wg  # noqa: E402
from bench import device_random  # noqa: E402

gpu = wg.GpuInstance.new(0)
dev, shapes = gpu.device(), wg.ViewShapeBuffers()
gemm, gemv = wg.Gemm.from_device(dev), wg.Gemv.from_device(dev)
S = wg.BufferUsages


def timed(fn, reps):
    def run(n):
        enc = dev.create_command_encoder()
        p = enc.compute_pass("t", None)
        for _ in range(n):
            fn(p)
        p.end()
        gpu.queue().submit([enc.finish()])
        gpu.sync()
    run(3)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        run(reps)
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6


def view(t, rows, cols, ld, off):
    return wg.GpuTensorView(wg.ViewShape((rows, cols, 1), ld, ld * cols, off), t, 2)


def gemm_case(dt, tr, M, N, K, pad=(0, 0, 0), off=(0, 0, 0)):
    """op(A) B with leading dimensions rows + pad and views starting `off` elements into their buffers."""
    ar, ac = (K, M) if tr else (M, K)
    lda, ldb, ldc = ar + pad[0], K + pad[1], M + pad[2]
    a = device_random(wg, gpu, (lda * ac + 8,), dt, 1)
    b = device_random(wg, gpu, (ldb * N + 8,), dt, 2)
    c = wg.TensorBuilder.vector(ldc * N + 8, S.STORAGE).build(dev, dt)
    av, bv, cv = view(a, ar, ac, lda, off[0]), view(b, K, N, ldb, off[1]), view(c, M, N, ldc, off[2])
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    reps = max(5, min(200, int(2e10 / max(1.0, 2.0 * M * N * K / (400.0 if dt == np.float16 else 100.0)) * 1e-6 * 50)))
    return timed(lambda p: gemm.dispatch_generic(dev, shapes, p, cv, av, bv, variant), reps)


def gemv_case(dt, tr, R, C, pad=0, off=(0, 0, 0)):
    ldm = R + pad
    m = device_random(wg, gpu, (ldm * C + 8,), dt, 1)
    vlen, olen = (R, C) if tr else (C, R)
    v = device_random(wg, gpu, (vlen + 8,), dt, 2)
    o = wg.TensorBuilder.vector(olen + 8, S.STORAGE).build(dev, dt)
    mv, vv, ov = view(m, R, C, ldm, off[0]), view(v, vlen, 1, vlen, off[1]), view(o, olen, 1, olen, off[2])
    variant = wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv
    return timed(lambda p: gemv.dispatch_generic(dev, shapes, p, ov, mv, vv, variant), 100)


def vec_case(dt, what, n, offs):
    a = device_random(wg, gpu, (n + 8,), dt, 1)
    b = device_random(wg, gpu, (n + 8,), dt, 2)
    res = wg.TensorBuilder.vector(4, S.STORAGE).build(dev, dt)
    av = wg.GpuTensorView(wg.ViewShape((n, 1, 1), n, n, offs[0]), a, 1)
    bv = wg.GpuTensorView(wg.ViewShape((n, 1, 1), n, n, offs[1]), b, 1)
    if what == "add":
        op = wg.OpAssign.new(dev, wg.OpAssignVariant.Add)
        return timed(lambda p: op.dispatch(dev, shapes, p, av, bv), 50)
    if what == "copy":
        op = wg.OpAssign.new(dev, wg.OpAssignVariant.Copy)
        return timed(lambda p: op.dispatch(dev, shapes, p, av, bv), 50)
    if what == "axpy":
        op = wg.Axpy.from_device(dev)
        return timed(lambda p: op.dispatch(dev, shapes, p, 0.5, av, bv), 50)
    red = wg.Reduce.new(dev, wg.ReduceOp.Sum)
    if what == "reduce_fast":
        return timed(lambda p: red.dispatch_fast(dev, shapes, p, av, res), 50)
    return timed(lambda p: red.dispatch(dev, shapes, p, av, res), 20)


for name in (sys.argv[1:] or ["f32", "f16"]):
    dt = np.float16 if name == "f16" else np.float32
    if os.environ.get("CLIFF_ONLY") in (None, "vec"):
        for what, n in [("add", 1 << 26), ("copy", 1 << 26), ("axpy", 1 << 26), ("reduce_fast", 1 << 26), ("reduce", 1 << 22)]:
            base = vec_case(dt, what, n, (0, 0))
            print(f"{name} {what} {n}: base {base:9.1f} us", flush=True)
            for label, offs in [("a+1", (1, 0)), ("b+1", (0, 1)), ("both+1", (1, 1)), ("a+4", (4, 0)), ("b+2", (0, 2))]:
                if what.startswith("reduce") and offs[0] == 0:
                    continue
                t = vec_case(dt, what, n, offs)
                print(f"    {label:8s} {t:9.1f} us  x{t / base:5.2f}{'   <-- cliff' if t > 1.3 * base else ''}", flush=True)
    if os.environ.get("CLIFF_ONLY") == "vec":
        continue
    for tr in (False, True):
        for (M, N, K) in ([] if os.environ.get("CLIFF_ONLY") == "gemv" else [(4096, 4096, 4096), (2048, 2048, 2048), (8192, 8192, 1024)] if dt == np.float16 else [(2048, 2048, 2048), (4096, 4096, 1024)]):
            base = gemm_case(dt, tr, M, N, K)
            print(f"{name} {'gemm_tr' if tr else 'gemm'} {M}x{N}x{K}: base {base:9.1f} us", flush=True)
            for label, kw in [("M+1", dict(dM=1)), ("M+4", dict(dM=4)), ("M+8", dict(dM=8)), ("N+1", dict(dN=1)), ("N+4", dict(dN=4)), ("N+8", dict(dN=8)), ("K+1", dict(dK=1)), ("K+4", dict(dK=4)),
                              ("K+8", dict(dK=8)), ("K+32", dict(dK=32)), ("lda+4", dict(pad=(4, 0, 0))), ("ldb+4", dict(pad=(0, 4, 0))), ("ldc+4", dict(pad=(0, 0, 4))), ("ld*+8", dict(pad=(8, 8, 8))),
                              ("offA 4", dict(off=(4, 0, 0))), ("offB 4", dict(off=(0, 4, 0))), ("offC 4", dict(off=(0, 0, 4))), ("offA 1", dict(off=(1, 0, 0))), ("offC 1", dict(off=(0, 0, 1)))]:
                t = gemm_case(dt, tr, M + kw.get("dM", 0), N + kw.get("dN", 0), K + kw.get("dK", 0), kw.get("pad", (0, 0, 0)), kw.get("off", (0, 0, 0)))
                print(f"    {label:8s} {t:9.1f} us  x{t / base:5.2f}{'   <-- cliff' if t > 1.3 * base else ''}", flush=True)
        for (R, C) in ([] if os.environ.get("CLIFF_ONLY") == "gemm" else [(8192, 8192), (4096, 16384), (16384, 16384)]):
            base = gemv_case(dt, tr, R, C)
            print(f"{name} {'gemv_tr' if tr else 'gemv'} {R}x{C}: base {base:9.1f} us", flush=True)
            for label, kw in [("R+1", dict(dR=1)), ("R+4", dict(dR=4)), ("C+1", dict(dC=1)), ("C+4", dict(dC=4)), ("ld+4", dict(pad=4)), ("ld+8", dict(pad=8)), ("offM 4", dict(off=(4, 0, 0))), ("offM 1", dict(off=(1, 0, 0))),
                              ("offV 1", dict(off=(0, 1, 0))), ("offV 4", dict(off=(0, 4, 0))), ("offO 1", dict(off=(0, 0, 1))), ("offO 4", dict(off=(0, 0, 4)))]:
                t = gemv_case(dt, tr, R + kw.get("dR", 0), C + kw.get("dC", 0), kw.get("pad", 0), kw.get("off", (0, 0, 0)))
                print(f"    {label:8s} {t:9.1f} us  x{t / base:5.2f}{'   <-- cliff' if t > 1.3 * base else ''}", flush=True)
