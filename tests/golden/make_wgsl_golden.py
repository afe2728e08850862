"""Generates tests/golden/wgsl_exec_*.npz by EXECUTING THE REFERENCE'S OWN SHADER TEXT with oracle/wgsl_exec.py.

    python tests/golden/make_wgsl_golden.py [/root/reference]

Needs the reference checkout (reads crates/wgebra/src/linalg/*.wgsl where they lie; nothing of it is copied into the
repo); the fixtures it writes are data only -- shapes, seeded inputs, the outputs the shaders produced -- and travel with the repo.
tests/test_oracle.py::test_restatement_matches_executed_wgsl then checks the NumPy and the C restatement against them BIT FOR BIT:
that pins the restatements against the reference's actual WGSL (entry points, index arithmetic, loop bounds, the lane/tree
reduction orders, the function redirection of reduce.rs / op_assign.rs) rather than against my reading of it.
What remains unpinned by this: the real wgpu/naga lowering of `mat4x4 * mat4x4` / `mat4x4 * vec4` (association and FMA contraction
are implementation-defined in WGSL; the executor and the restatements both use left-to-right, unfused) -- see oracle/wgsl_exec.py.

Dispatch grids follow the Rust side: gemm.rs:109-126, gemv.rs:99-136, reduce.rs:110-112, op_assign.rs:92-94.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import wgsl_exec as wx  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
LINALG = os.path.join(REF, "crates", "wgebra", "src", "linalg")
FIELDS = ["nrows", "ncols", "nmats", "stride", "stride_mat", "offset"]


def shape(nrows, ncols=1, nmats=1, stride=None, stride_mat=None, offset=0):
    stride = nrows if stride is None else stride
    stride_mat = nrows * ncols if stride_mat is None else stride_mat
    return dict(zip(FIELDS, (nrows, ncols, nmats, stride, stride_mat, offset)))


def sv(s):
    return np.array([s[f] for f in FIELDS], np.int64)


def upm(rng, n):
    return (rng.random(n, dtype=np.float32) * np.float32(2) - np.float32(1)).astype(np.float32)


def div_ceil(a, b):
    return -(-a // b)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def gemm_cases():
    mod = wx.load_linalg(LINALG, "gemm.wgsl")
    out = {}
    # (M, K, N, mats): fast variants need K % 256 == 0 (64 lanes x 4 columns per trip, gemm.wgsl:39-40)
    cases = [(8, 256, 8, 1), (64, 256, 12, 2), (20, 512, 4, 1)]
    for ci, (M, K, N, mats) in enumerate(cases):
        rng = np.random.default_rng(100 + ci)
        for tr in (0, 1):
            s1 = shape(K, M, mats) if tr else shape(M, K, mats)
            s2, so = shape(K, N, mats), shape(M, N, mats)
            m1, m2 = upm(rng, M * K * mats), upm(rng, K * N * mats)
            for fast in (0, 1):
                entry = ("gemm_tr" if tr else "gemm") + ("_fast" if fast else "")
                o = np.full(M * N * mats, np.nan, np.float32)
                gx = div_ceil(M, 4) if fast else div_ceil(M, 64)  # gemm.rs:109-116
                mod.run(entry, (gx, mats, 1), {"shape_out": so, "shape_m1": s1, "shape_m2": s2, "out": o, "m1": m1, "m2": m2})
                k = f"c{ci}_{entry}"
                out[k + "_m1"], out[k + "_m2"], out[k + "_out"] = m1, m2, o
                out[k + "_shapes"] = np.stack([sv(so), sv(s1), sv(s2)])
    # a strided / offset view case (naive variants): parent 24 x K, rows 4..19 of columns, offset multiple of 4
    M, K, N = 16, 64, 8
    rng = np.random.default_rng(150)
    pm1 = upm(rng, 24 * K + 16)
    s1 = shape(M, K, 1, stride=24, stride_mat=24 * K, offset=4)
    s2 = shape(K, N)
    so = shape(M, N, 1, stride=20, stride_mat=20 * N, offset=8)
    m2, o = upm(rng, K * N), np.full(8 + 20 * N, np.nan, np.float32)
    mod.run("gemm", (div_ceil(M, 64), 1, 1), {"shape_out": so, "shape_m1": s1, "shape_m2": s2, "out": o, "m1": pm1, "m2": m2})
    out["view_gemm_m1"], out["view_gemm_m2"], out["view_gemm_out"] = pm1, m2, o
    out["view_gemm_shapes"] = np.stack([sv(so), sv(s1), sv(s2)])
    save("wgsl_exec_gemm", **out)


def gemv_cases():
    mod = wx.load_linalg(LINALG, "gemv.wgsl")
    out = {}
    cases = [(8, 128, 1, 1), (64, 256, 3, 2), (128, 128, 2, 1), (20, 384, 1, 1)]  # (R, C, nrhs, mats)
    for ci, (R, C, nrhs, mats) in enumerate(cases):
        rng = np.random.default_rng(200 + ci)
        m = upm(rng, R * C * mats)
        sm = shape(R, C, mats)
        for tr in (0, 1):
            vlen, olen = (R, C) if tr else (C, R)
            v = upm(rng, vlen * nrhs * mats)
            sv_, so = shape(vlen, nrhs, mats), shape(olen, nrhs, mats)
            for fast in (0, 1):
                entry = ("gemv_tr" if tr else "gemv") + ("_fast" if fast else "")
                if fast and not tr and C % 128:          # gemv_fast walks 32 lanes x 4 columns per trip (gemv.wgsl:40-41)
                    continue
                if fast and tr and R % 128:              # gemv.rs:99-104: GemvTrFast falls back to GemvTr
                    continue
                if fast and olen % 4:                    # gemv.rs:122
                    continue
                o = np.full(olen * nrhs * mats, np.nan, np.float32)
                gx = div_ceil(olen, 4) if fast else div_ceil(olen, 32)  # gemv.rs:113-126
                mod.run(entry, (gx, nrhs, mats), {"shape_out": so, "shape_m": sm, "shape_v": sv_, "out": o, "m": m, "v": v})
                k = f"c{ci}_{entry}"
                out[k + "_m"], out[k + "_v"], out[k + "_out"] = m, v, o
                out[k + "_shapes"] = np.stack([sv(so), sv(sm), sv(sv_)])
    save("wgsl_exec_gemv", **out)


RED_OPS = {  # reduce.rs:29-58: (init_fn, workspace_fn, reduce_fn)
    0: ("init_max_f32", "reduce_min_f32", "reduce_min_f32"), 1: ("init_min_f32", "reduce_max_f32", "reduce_max_f32"),
    2: ("init_zero", "reduce_sum_f32", "reduce_sum_f32"), 3: ("init_one", "reduce_prod_f32", "reduce_prod_f32"),
    4: ("init_zero", "reduce_sqnorm_f32", "reduce_sum_f32")}


def reduce_cases():
    out = {}
    for op, (init, ws, red) in RED_OPS.items():
        mod = wx.load_linalg(LINALG, "reduce.wgsl", redirect={"init_placeholder": init, "workspace_placeholder": ws, "reduce_placeholder": red})
        for n, off in [(0, 0), (1, 0), (127, 0), (128, 0), (129, 3), (345, 0), (1000, 5)]:
            rng = np.random.default_rng(300 + n)
            x = np.concatenate([np.zeros(off, np.float32), upm(rng, n) if op != 3 else (np.float32(0.5) + rng.random(n, dtype=np.float32))]).astype(np.float32)
            if x.size == 0:
                x = np.zeros(1, np.float32)
            res = np.full(1, np.nan, np.float32)
            mod.run("main", (1, 1, 1), {"shape": shape(n, 1, 1, n, n, off), "input": x, "output": res})
            k = f"op{op}_n{n}_o{off}"
            out[k + "_x"], out[k + "_res"] = x, res
    save("wgsl_exec_reduce", **out)


OPA = {0: "add_f32", 1: "sub_f32", 2: "mul_f32", 3: "div_f32", 4: "copy_f32"}  # op_assign.rs:28-38


def op_assign_cases():
    out = {}
    for op, fn in OPA.items():
        mod = wx.load_linalg(LINALG, "op_assign.wgsl", redirect={"placeholder": fn})
        for n, oa, ob in [(1757, 0, 0), (100, 3, 7), (64, 0, 1)]:
            rng = np.random.default_rng(400 + n + op)
            a = upm(rng, n + oa)
            b = (np.float32(0.25) + rng.random(n + ob, dtype=np.float32)).astype(np.float32)
            a0 = a.copy()
            mod.run("main", (div_ceil(n, 64), 1, 1), {"shape_a": shape(n, 1, 1, n, n, oa), "shape_b": shape(n, 1, 1, n, n, ob), "a": a, "b": b})
            k = f"op{op}_n{n}_{oa}_{ob}"
            out[k + "_a0"], out[k + "_b"], out[k + "_a"] = a0, b, a
    save("wgsl_exec_op_assign", **out)


def shape_cases():
    """shape.wgsl's index functions, column-major and with the ROW_MAJOR definition (shape.wgsl:36-66), on random shapes / indices."""
    out = {}
    for tag, defs in (("cm", set()), ("rm", {"ROW_MAJOR"})):
        mod = wx.Module(open(os.path.join(LINALG, "shape.wgsl")).read(), defs=defs)
        rng = np.random.default_rng(500)
        rows = []
        for _ in range(64):
            nrows, ncols, nmats = (int(x) for x in rng.integers(1, 200, 3))
            stride, stride_mat, offset = int(rng.integers(1, 400)), int(rng.integers(1, 100000)), int(rng.integers(0, 1000))
            i, j, t = int(rng.integers(0, nrows)), int(rng.integers(0, ncols)), int(rng.integers(0, nmats))
            sh = wx.Struct(FIELDS, [nrows, ncols, nmats, stride, stride_mat, offset])

            def call(name, *args):
                g = mod.ns["F_" + name](None, *args)
                try:
                    while True:
                        next(g)
                except StopIteration as e:
                    return e.value
            v4 = call("with_vec4_elts", sh)
            rows.append([nrows, ncols, nmats, stride, stride_mat, offset, i, j, t, call("it", sh, i, j, t), call("iv", sh, i),
                         v4.nrows, v4.ncols, v4.nmats, v4.stride, v4.stride_mat, v4.offset])
        out[tag] = np.array(rows, np.int64)
    save("wgsl_exec_shape", **out)


if __name__ == "__main__":
    shape_cases()
    op_assign_cases()
    reduce_cases()
    gemv_cases()
    gemm_cases()
