"""One-off stress of round 6's any-alignment paths: random Gemm / Gemv / CopyView calls on views with random (odd) offsets, leading dimensions, batch strides and lengths,
f32 and f16, against f64 / NumPy; nothing outside an output view may change. Usage (GPU box): python tools/archive/r06/fuzz_alignment.py [cases] [seed]"""
import os
import sys

root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np

import wgmath_amd as wg
from wgmath_amd import _lib

gpu = wg.GpuInstance.new(0)
dev, shapes = gpu.device(), wg.ViewShapeBuffers()
S = wg.BufferUsages.STORAGE | wg.BufferUsages.COPY_SRC | wg.BufferUsages.COPY_DST
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261003)


def up(flat, dt):
    return wg.TensorBuilder.tensor((flat.size,), S).build_init(dev, np.asarray(flat, dt), dt)


def run(fn):
    enc = dev.create_command_encoder()
    with enc.compute_pass("fuzz", None) as p:
        fn(p)
    gpu.queue().submit([enc.finish()])


def dim(lo, hi, mult_p=0.5):
    v = int(rng.integers(lo, hi + 1))
    if rng.random() < mult_p:
        v = max(lo, (v // 8) * 8) or 8
    return max(1, v)


bad = 0
gemm, gemv = wg.Gemm.from_device(dev), wg.Gemv.from_device(dev)
for case in range(ncases):
    dt = np.float16 if rng.random() < 0.5 else np.float32
    eps = 2.0 ** -10 if dt == np.float16 else 2.0 ** -22
    kind = rng.choice(["gemm", "gemm", "gemv", "copy"])
    tr = bool(rng.random() < 0.5)
    mats = int(rng.choice([1, 1, 1, 2, 3]))
    if kind == "gemm":
        big = rng.random() < 0.3
        M, N, K = dim(1, 2300 if big else 300), dim(1, 2300 if big else 300), dim(1, 1500 if big else 400)
        ar, ac = (K, M) if tr else (M, K)
        pa, pb, pc = (int(x) for x in rng.integers(0, 9, 3))
        oa, ob, oc = (int(x) for x in rng.integers(0, 9, 3))
        ba, bb, bc = (int(x) for x in rng.integers(0, 5, 3))
        lda, ldb, ldc = ar + pa, K + pb, M + pc
        sa, sb, sc = lda * ac + ba, ldb * N + bb, ldc * N + bc
        A = (rng.random((mats, ac, ar), dtype=np.float32) - 0.5).astype(dt)
        B = (rng.random((mats, N, K), dtype=np.float32) - 0.5).astype(dt)
        fa = np.zeros(oa + sa * mats + 8, dt); fb = np.zeros(ob + sb * mats + 8, dt)
        fc = rng.random(oc + sc * mats + 8, dtype=np.float32).astype(dt)
        for z in range(mats):
            fa[oa + z * sa: oa + z * sa + lda * ac].reshape(ac, lda)[:, :ar] = A[z]
            fb[ob + z * sb: ob + z * sb + ldb * N].reshape(N, ldb)[:, :K] = B[z]
        ta, tb, tc = up(fa, dt), up(fb, dt), up(fc, dt)
        av = wg.GpuTensorView(wg.ViewShape((ar, ac, mats), lda, sa, oa), ta, 2)
        bv = wg.GpuTensorView(wg.ViewShape((K, N, mats), ldb, sb, ob), tb, 2)
        cv = wg.GpuTensorView(wg.ViewShape((M, N, mats), ldc, sc, oc), tc, 2)
        run(lambda p: gemm.dispatch_generic(dev, shapes, p, cv, av, bv, wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm))
        got = tc.read(dev)
        mask = np.ones(fc.size, bool)
        ok = True
        for z in range(mats):
            A64 = A[z].astype(np.float64).T if not tr else A[z].astype(np.float64)
            B64 = B[z].astype(np.float64).T
            ref, sabs = A64 @ B64, np.abs(A64) @ np.abs(B64)
            idx = oc + z * sc + np.arange(M)[:, None] + np.arange(N)[None, :] * ldc
            G = got[idx].astype(np.float64)
            tol = (K + 8) * 2.0 ** -22 * sabs + eps * np.abs(ref) + (2.0 ** -24 if dt == np.float16 else 1e-30)  # (f16 results below 2^-14 are subnormal: half a unit of 2^-24)
            ok &= bool((np.abs(G - ref) <= tol).all())
            mask[idx.ravel()] = False
        ok &= bool(np.array_equal(got[mask].view(np.uint8), fc[mask].view(np.uint8)))
        desc = f"gemm {np.dtype(dt).name} tr={tr} M{M} N{N} K{K} x{mats} pads {(pa, pb, pc)} offs {(oa, ob, oc)} batch {(ba, bb, bc)}"
    elif kind == "gemv":
        R, C = dim(1, 3000), dim(1, 3000)
        n = int(rng.choice([1, 1, 2, 3, 5, 8]))
        ld = R + int(rng.integers(0, 9))
        om, ov, oo = (int(x) for x in rng.integers(0, 9, 3))
        vlen, olen = (R, C) if tr else (C, R)
        ldv, ldo = vlen + int(rng.integers(0, 5)), olen + int(rng.integers(0, 5))
        pm = (rng.random(om + ld * C * mats + 8, dtype=np.float32) - 0.5).astype(dt)
        pv = (rng.random(ov + ldv * n * mats + 8, dtype=np.float32) - 0.5).astype(dt)
        po = rng.random(oo + ldo * n * mats + 8, dtype=np.float32).astype(dt)
        tm, tv, to = up(pm, dt), up(pv, dt), up(po, dt)
        mv = wg.GpuTensorView(wg.ViewShape((R, C, mats), ld, ld * C, om), tm, 2)
        vv = wg.GpuTensorView(wg.ViewShape((vlen, n, mats), ldv, ldv * n, ov), tv, 2)
        ovw = wg.GpuTensorView(wg.ViewShape((olen, n, mats), ldo, ldo * n, oo), to, 2)
        run(lambda p: gemv.dispatch_generic(dev, shapes, p, ovw, mv, vv, wg.GemvVariant.GemvTr if tr else wg.GemvVariant.Gemv))
        got = to.read(dev)
        mask = np.ones(po.size, bool)
        ok = True
        for z in range(mats):
            A = pm[om + z * ld * C: om + z * ld * C + ld * C].reshape(C, ld)[:, :R].T.astype(np.float64)
            X = pv[ov + z * ldv * n: ov + (z + 1) * ldv * n].reshape(n, ldv)[:, :vlen].T.astype(np.float64)
            A = A.T if tr else A
            ref, sabs = A @ X, np.abs(A) @ np.abs(X)
            idx = oo + z * ldo * n + np.arange(olen)[:, None] + np.arange(n)[None, :] * ldo
            G = got[idx].astype(np.float64)
            tol = (vlen + 8) * 2.0 ** -22 * sabs + eps * np.abs(ref) + (2.0 ** -24 if dt == np.float16 else 1e-30)
            ok &= bool((np.abs(G - ref) <= tol).all())
            mask[idx.ravel()] = False
        ok &= bool(np.array_equal(got[mask].view(np.uint8), po[mask].view(np.uint8)))
        desc = f"gemv {np.dtype(dt).name} tr={tr} R{R} C{C} ld{ld} n{n} x{mats} offs {(om, ov, oo)}"
    else:
        rs, cs, rd, cd = dim(1, 3000), dim(1, 40), dim(1, 3000), dim(1, 40)
        lds, ldd = rs + int(rng.integers(0, 9)), rd + int(rng.integers(0, 9))
        so, do = int(rng.integers(0, 9)), int(rng.integers(0, 9))
        src = (rng.random(so + lds * cs * mats + 16, dtype=np.float32) - 0.5).astype(dt)
        dst0 = (rng.random(do + ldd * cd * mats + 16, dtype=np.float32) + 1.0).astype(dt)
        ts, td = up(src, dt), up(dst0, dt)
        _lib.check(_lib.lib.wg_copy_view(gpu._ctx.handle, wg.wgcore.wg_dtype(dt), td._h, wg.ViewShape((rd, cd, mats), ldd, ldd * cd, do).to_c(), ts._h,
                                         wg.ViewShape((rs, cs, mats), lds, lds * cs, so).to_c()))
        got = td.read(dev)
        want = dst0.copy()
        for z in range(mats):
            for j in range(cd):
                col = np.zeros(rd, dt)
                if j < cs:
                    k = min(rs, rd)
                    col[:k] = src[so + z * lds * cs + j * lds: so + z * lds * cs + j * lds + k]
                want[do + z * ldd * cd + j * ldd: do + z * ldd * cd + j * ldd + rd] = col
        ok = bool(np.array_equal(got.view(np.uint8), want.view(np.uint8)))
        desc = f"copy {np.dtype(dt).name} src {rs}x{cs} ld{lds}+{so} -> dst {rd}x{cd} ld{ldd}+{do} x{mats}"
    if not ok:
        bad += 1
        print("FAIL", desc, flush=True)
print(f"{ncases} random cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
