"""Generates tests/golden/*.npz: seeded inputs + the outputs of the CPU restatement (oracle/), so that the GPU parity
tests (and the oracle's own regression tests) have committed vectors that do not depend on the generator's RNG.

    python tests/golden/make_golden.py

The reference holds NO golden vectors for Gemm/Gemv/Reduce (its tests draw unseeded random inputs and compare with
nalgebra, abs eps 1e-3) and cannot be executed here (Rust + wgpu), so these are vectors of the restatement, not of a
WGSL run: they pin the restatement against drift and give the HIP kernels fixed inputs.  The one deterministic test the
reference does hold -- gpu_op_assign (op_assign.rs:110-155: v0[i] = i + 0.1, v1[i] = 10 i + 0.1, LEN 1757) -- is
reproduced exactly; its expected values are IEEE f32 + - * / of those inputs.

Every expected array is produced by the NumPy restatement and asserted bit-identical to the C restatement before it is
written.  f64 ground truth (`truth`) and sum|a||b| (`sabs`) are stored for the tolerance checks.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import wgsl_oracle as wo  # noqa: E402

C = wo.CLib()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def u01(rng, n):
    return rng.random(n, dtype=np.float32)


def upm(rng, n):
    return (rng.random(n, dtype=np.float32) * np.float32(2) - np.float32(1)).astype(np.float32)


def gemm_case(seed, M, K, N, mats, signed):
    rng = np.random.default_rng(seed)
    gen = upm if signed else u01
    out = {}
    for tr in (0, 1):
        s1 = wo.Shape(K, M, mats) if tr else wo.Shape(M, K, mats)
        s2, so = wo.Shape(K, N, mats), wo.Shape(M, N, mats)
        m1, m2 = gen(rng, M * K * mats), gen(rng, K * N * mats)
        key = "tr" if tr else "nn"
        out[f"m1_{key}"], out[f"m2_{key}"] = m1, m2
        for fast in (0, 1):
            variant = (wo.GEMM_TR if tr else wo.GEMM) + fast
            o_np, o_c = np.zeros(M * N * mats, np.float32), np.zeros(M * N * mats, np.float32)
            wo.gemm(variant, o_np, so, m1, s1, m2, s2)
            C.gemm(variant, o_c, so, m1, s1, m2, s2)
            assert o_np.tobytes() == o_c.tobytes()
            out[f"out_v{variant}"] = o_np
        a = wo.view(m1, s1)
        b = wo.view(m2, s2)
        truth = np.empty((M, N, mats)); sabs = np.empty((M, N, mats))
        for t in range(mats):
            amk = a[:, :, t].T if tr else a[:, :, t]
            truth[:, :, t], sabs[:, :, t] = wo.gemm_f64(amk, b[:, :, t])
        out[f"truth_{key}"] = truth.reshape(-1, order="F")
        out[f"sabs_{key}"] = sabs.reshape(-1, order="F").astype(np.float32)
    out["dims"] = np.array([M, K, N, mats], np.int64)
    return out


def gemv_case(seed, R, Cc, nrhs, mats, signed):
    rng = np.random.default_rng(seed)
    gen = upm if signed else u01
    out = {"dims": np.array([R, Cc, nrhs, mats], np.int64)}
    m = gen(rng, R * Cc * mats)
    out["m"] = m
    sm = wo.Shape(R, Cc, mats)
    for tr in (0, 1):
        vlen, olen = (R, Cc) if tr else (Cc, R)
        v = gen(rng, vlen * nrhs * mats)
        key = "tr" if tr else "nn"
        out[f"v_{key}"] = v
        sv, so = wo.Shape(vlen, nrhs, mats), wo.Shape(olen, nrhs, mats)
        for fast in (0, 1):
            variant = (wo.GEMV_TR if tr else wo.GEMV) + fast
            o_np = gen(rng, olen * nrhs * mats)  # pre-filled with noise: the kernel must overwrite (gemv.rs:163,169-170)
            o_c = o_np.copy()
            wo.gemv(variant, o_np, so, m, sm, v, sv)
            C.gemv(variant, o_c, so, m, sm, v, sv)
            assert o_np.tobytes() == o_c.tobytes()
            out[f"out_v{variant}"] = o_np
        a, x = wo.view(m, sm), wo.view(v, sv)
        truth = np.empty((olen, nrhs, mats)); sabs = np.empty((olen, nrhs, mats))
        for t in range(mats):
            amk = a[:, :, t].T if tr else a[:, :, t]
            truth[:, :, t], sabs[:, :, t] = wo.gemm_f64(amk, x[:, :, t])
        out[f"truth_{key}"] = truth.reshape(-1, order="F")
        out[f"sabs_{key}"] = sabs.reshape(-1, order="F").astype(np.float32)
    return out


def main():
    # GEMM: 64 x 256 x 32, 2 matrices, all four variants (K = 256 is the smallest K the *_fast kernels accept)
    save("gemm_u01_64x256x32x2", **gemm_case(0xC0FFEE + 2, 64, 256, 32, 2, signed=False))
    save("gemm_pm1_64x256x32x2", **gemm_case(0xC0FFEE + 3, 64, 256, 32, 2, signed=True))
    # GEMV: 128 x 256, 3 RHS columns x 2 matrices, all four variants
    save("gemv_u01_128x256x3x2", **gemv_case(0xC0FFEE + 4, 128, 256, 3, 2, signed=False))
    save("gemv_pm1_128x256x3x2", **gemv_case(0xC0FFEE + 5, 128, 256, 3, 2, signed=True))

    # Reduce: n in {0, 1, 127, 128, 129, 345, 65536} x 5 ops, bit-exact expectations
    rng = np.random.default_rng(0xC0FFEE + 6)
    red = {}
    for n in (0, 1, 127, 128, 129, 345, 65536):
        x = u01(rng, n) if n != 65536 else upm(rng, n)
        if n == 345:  # the reference's own length (reduce.rs:151), its distribution U[0,1)
            x = u01(rng, n)
        red[f"x_{n}"] = x
        exp = np.empty(5, np.float32)
        for op in range(5):
            r_np = wo.reduce(op, x, wo.Shape(n))
            if n > 0:
                r_c = C.reduce(op, x, wo.Shape(n))
                assert np.float32(r_np).tobytes() == np.float32(r_c).tobytes(), (n, op)
            exp[op] = r_np
        red[f"expected_{n}"] = exp
    # Prod on U[0,1) underflows for long vectors; a near-1 vector keeps the product informative
    xp = (np.float32(1) + (rng.random(4096, dtype=np.float32) - np.float32(0.5)) * np.float32(1e-2)).astype(np.float32)
    red["x_prod4096"] = xp
    red["expected_prod4096"] = np.array([wo.reduce(op, xp, wo.Shape(4096)) for op in range(5)], np.float32)
    # batched: 96 columns of 1000 (ragged last row), stride 1000, offset 0 -> aligned fast path; and an odd stride
    xb = upm(rng, 1000 * 96)
    red["xb"] = xb
    for op in range(5):
        e = wo.reduce_batched(op, xb, wo.Shape(1000, 96))
        assert e.tobytes() == C.reduce_batched(op, xb, wo.Shape(1000, 96)).tobytes()
        red[f"expected_batched_{op}"] = e
    save("reduce", **red)

    # OpAssign: the reference's deterministic vectors (op_assign.rs:123-126)
    LEN = 1757
    v0 = (np.arange(LEN, dtype=np.float32) + np.float32(0.1)).astype(np.float32)
    v1 = (np.arange(LEN, dtype=np.float32) * np.float32(10.0) + np.float32(0.1)).astype(np.float32)
    oa = {"v0": v0, "v1": v1}
    for op in range(5):
        a_np, a_c = v0.copy(), v0.copy()
        wo.op_assign(op, a_np, wo.Shape(LEN), v1, wo.Shape(LEN))
        C.op_assign(op, a_c, wo.Shape(LEN), v1, wo.Shape(LEN))
        assert a_np.tobytes() == a_c.tobytes()
        oa[f"expected_{op}"] = a_np
    save("op_assign_ref_1757", **oa)


if __name__ == "__main__":
    main()
