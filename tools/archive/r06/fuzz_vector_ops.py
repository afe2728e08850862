"""One-off stress of OpAssign / Axpy / Reduce (Min, Max) on vector views at random offsets and lengths, f32 and f16, against NumPy bit for bit (IEEE + - * / and the one
rounding of the f16 forms are exact operations; Min / Max are order-free); nothing outside the written view may change. Usage (GPU box): python tools/archive/r06/fuzz_vector_ops.py [cases] [seed]"""
import os
import sys

root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
import numpy as np

import wgmath_amd as wg

gpu = wg.GpuInstance.new(0)
dev, shapes = gpu.device(), wg.ViewShapeBuffers()
S = wg.BufferUsages.STORAGE | wg.BufferUsages.COPY_SRC | wg.BufferUsages.COPY_DST
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)


def up(flat, dt):
    return wg.TensorBuilder.tensor((flat.size,), S).build_init(dev, np.asarray(flat, dt), dt)


def run(fn):
    enc = dev.create_command_encoder()
    with enc.compute_pass("fuzz", None) as p:
        fn(p)
    gpu.queue().submit([enc.finish()])


bad = 0
for case in range(ncases):
    dt = np.float16 if rng.random() < 0.5 else np.float32
    n = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 31, 33, 100, 1000, 4097, 65537, 1 << 20, (1 << 20) + 3, int(rng.integers(1, 300000))]))
    oa, ob = int(rng.integers(0, 17)), int(rng.integers(0, 17))
    a = ((rng.random(n + oa + 9, dtype=np.float32) - 0.5) * 8).astype(dt)
    b = ((rng.random(n + ob + 9, dtype=np.float32) - 0.3) * 4 + 0.01).astype(dt)
    ta, tb = up(a, dt), up(b, dt)
    av = wg.GpuTensorView(wg.ViewShape((n, 1, 1), n, n, oa), ta, 1)
    bv = wg.GpuTensorView(wg.ViewShape((n, 1, 1), n, n, ob), tb, 1)
    kind = str(rng.choice(["add", "sub", "mul", "div", "copy", "axpy", "min", "max"]))
    A, B = a[oa:oa + n], b[ob:ob + n]
    if kind in ("min", "max"):
        res = wg.TensorBuilder.vector(1, S).build(dev, dt)
        op = wg.Reduce.new(dev, wg.ReduceOp.Min if kind == "min" else wg.ReduceOp.Max)
        fast = bool(rng.random() < 0.5)
        run(lambda p: (op.dispatch_fast if fast else op.dispatch)(dev, shapes, p, av, res))
        got = res.read(dev)[0]
        want = A.min() if kind == "min" else A.max()
        ok = got.tobytes() == np.asarray(want, dt).tobytes()
        desc = f"reduce {kind}{' fast' if fast else ''} {np.dtype(dt).name} n {n} offset {oa}"
    else:
        want = a.copy()
        with np.errstate(all="ignore"):
            if kind == "axpy":
                alpha = np.float32(rng.choice([0.5, -2.0, 1.0, 0.3]))
                r32 = (np.float64(alpha) * B.astype(np.float64) + A.astype(np.float64)).astype(np.float32)  # one rounding of the exact fma (f64 holds it exactly for these operands)
                want[oa:oa + n] = r32.astype(dt)
                run(lambda p: wg.Axpy.from_device(dev).dispatch(dev, shapes, p, float(alpha), av, bv))
            else:
                f = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide, "copy": lambda x, y: y}[kind]
                want[oa:oa + n] = f(A.astype(np.float32), B.astype(np.float32)).astype(dt) if dt == np.float16 else f(A, B)
                v = {"add": wg.OpAssignVariant.Add, "sub": wg.OpAssignVariant.Sub, "mul": wg.OpAssignVariant.Mul, "div": wg.OpAssignVariant.Div, "copy": wg.OpAssignVariant.Copy}[kind]
                run(lambda p: wg.OpAssign.new(dev, v).dispatch(dev, shapes, p, av, bv))
        got = ta.read(dev)
        ok = bool(np.array_equal(got.view(np.uint8), want.view(np.uint8)))
        desc = f"{kind} {np.dtype(dt).name} n {n} offsets {(oa, ob)}"
    if not ok:
        bad += 1
        print("FAIL", desc, flush=True)
print(f"{ncases} random cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
