"""Randomised end-to-end check of Gemm / Gemv (both variants, f32 and f16, batches, several right-hand sides, padded leading dimensions, offsets)
against f64 on the host, for a time budget: python tools/fuzz_gpu.py [seconds] [seed].  Shapes are drawn around the launchers' decision boundaries
(tile counts near the CU count, K remainders, few rows / columns, 3-8 right-hand sides around the size thresholds). Prints every failure; exit code 1 if any."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

import wgmath_amd as wg  # noqa: E402
import _util as U  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) & 0xffff
rng = np.random.default_rng(seed)
gpu = wg.GpuInstance.new(0)
dev, shapes = gpu.device(), wg.ViewShapeBuffers()
S = wg.BufferUsages
fails, runs = 0, 0


def up(arr, dt):
    return wg.TensorBuilder.vector(arr.size, S.STORAGE | S.COPY_SRC | S.COPY_DST).build_init(dev, np.ascontiguousarray(arr.reshape(-1)).astype(dt))


def pick(cands):
    return int(cands[rng.integers(0, len(cands))])


def dim(align, big):
    base = pick([1, 2, 3, 5, 8, 16, 31, 32, 33, 64, 100, 127, 128, 129, 250, 256, 257, 300, 511, 512, 520, 1000, 1024, 1030, 2048] + ([2600, 4096, 4100] if big else []))
    return max(align, (base * pick([1, 1, 1, 2]) // align) * align)


def check(got, truth, sabs, k, f16, what):
    global fails
    tol = U.f32_gate(k, sabs)
    if f16:
        tol = tol + 2.0 ** -11 * np.abs(truth) + 2.0 ** -25
    err = np.abs(got - truth)
    if not (err <= tol).all():
        fails += 1
        print(f"FAIL {what}: worst err/tol {(err / tol).max():.3g}", flush=True)


t_end = time.time() + budget
while time.time() < t_end:
    runs += 1
    f16 = bool(rng.integers(0, 2))
    dt = np.float16 if f16 else np.float32
    tr = bool(rng.integers(0, 2))
    al = 8 if f16 else 4
    kind = pick([0, 0, 1])
    if kind == 0:  # Gemm
        M, N, K = dim(al, True), dim(4, True), dim(al if f16 else 4, True)
        if M * N * K > 6e9:
            continue
        mats = pick([1, 1, 1, 2, 3]) if M * N * K < 5e8 else 1
        pa, pb, pc = pick([0, 0, al]), pick([0, 0, al]), pick([0, 0, al])  # leading-dimension padding
        ar, ac = (K, M) if tr else (M, K)
        a = (rng.random((mats, ac, ar + pa), dtype=np.float32) * 2 - 1).astype(dt)
        b = (rng.random((mats, N, K + pb), dtype=np.float32) * 2 - 1).astype(dt)
        c0 = np.full((mats, N, M + pc), np.nan, dt)
        ta, tb, tc = up(a, dt), up(b, dt), up(c0, dt)
        va = wg.GpuTensorView(wg.ViewShape((ar, ac, mats), ar + pa, (ar + pa) * ac, 0), ta, 3)
        vb = wg.GpuTensorView(wg.ViewShape((K, N, mats), K + pb, (K + pb) * N, 0), tb, 3)
        vc = wg.GpuTensorView(wg.ViewShape((M, N, mats), M + pc, (M + pc) * N, 0), tc, 3)
        gemm = wg.Gemm.from_device(dev)
        enc = dev.create_command_encoder()
        p = enc.compute_pass("f", None)
        try:
            gemm.dispatch_generic(dev, shapes, p, vc, va, vb, wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print(f"ERROR gemm {np.dtype(dt).name} tr={tr} {M}x{N}x{K} x{mats} pads {pa},{pb},{pc}: {e}", flush=True)
            continue
        p.end()
        gpu.queue().submit([enc.finish()])
        got = tc.read(dev).reshape(mats, N, M + pc)[:, :, :M].astype(np.float64)
        for z in range(mats):
            A = a[z, :, :ar].astype(np.float64)
            A = A if tr else A.T  # op(A): M x K
            A = A.T if tr else A
            Bm = b[z, :, :K].astype(np.float64).T  # K x N
            opA = (a[z, :, :ar].astype(np.float64)).T if not tr else a[z, :, :ar].astype(np.float64)
            # a[z] is [col][row] of the stored matrix: stored (ar x ac). Gemm: stored M x K -> opA = stored; GemmTr: stored K x M -> opA = stored^T
            stored = a[z, :, :ar].astype(np.float64).T  # ar x ac
            opA = stored.T if tr else stored
            truth, sabs = opA @ Bm, np.abs(opA) @ np.abs(Bm)
            check(got[z].T, truth, sabs, K, f16, f"gemm {np.dtype(dt).name} tr={tr} {M}x{N}x{K} mat {z}/{mats} pads {pa},{pb},{pc}")
    else:  # Gemv
        R, C = dim(al, True), dim(al, True)
        nrhs = pick([1, 1, 2, 3, 4, 5, 8, 9, 12])
        if nrhs > 1 and rng.integers(0, 2):
            R, C = pick([2048, 4096, 4104, 6144]), pick([2048, 3072, 4096, 6152])  # around the few-right-hand-sides thresholds
        mats = pick([1, 1, 2]) if R * C < 2e6 else 1
        m = (rng.random((mats, C, R), dtype=np.float32) * 2 - 1).astype(dt)
        vlen, olen = (R, C) if tr else (C, R)
        v = (rng.random((mats, nrhs, vlen), dtype=np.float32) * 2 - 1).astype(dt)
        tm, tv = up(m, dt), up(v, dt)
        to = up(np.full((mats, nrhs, olen), np.nan, dt), dt)
        vm = wg.GpuTensorView(wg.ViewShape((R, C, mats), R, R * C, 0), tm, 3)
        vv = wg.GpuTensorView(wg.ViewShape((vlen, nrhs, mats), vlen, vlen * nrhs, 0), tv, 3)
        vo = wg.GpuTensorView(wg.ViewShape((olen, nrhs, mats), olen, olen * nrhs, 0), to, 3)
        gemv = wg.Gemv.from_device(dev)
        enc = dev.create_command_encoder()
        p = enc.compute_pass("f", None)
        try:
            gemv.dispatch_generic(dev, shapes, p, vo, vm, vv, wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print(f"ERROR gemv {np.dtype(dt).name} tr={tr} {R}x{C} rhs={nrhs} x{mats}: {e}", flush=True)
            continue
        p.end()
        gpu.queue().submit([enc.finish()])
        got = to.read(dev).reshape(mats, nrhs, olen).astype(np.float64)
        for z in range(mats):
            M64 = m[z].astype(np.float64).T  # R x C
            op = M64.T if tr else M64
            X = v[z].astype(np.float64).T  # vlen x nrhs
            truth, sabs = op @ X, np.abs(op) @ np.abs(X)
            check(got[z].T, truth, sabs, vlen, f16, f"gemv {np.dtype(dt).name} tr={tr} {R}x{C} rhs={nrhs} mat {z}/{mats}")
print(f"fuzz: {runs} cases, {fails} failures, seed {seed}")
sys.exit(1 if fails else 0)
