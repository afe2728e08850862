"""Shared helpers for the parity tests: tolerances (stated once, here) and fixture loading."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# ---- tolerances ---------------------------------------------------------------------------------------------
# OpAssign (+ - * / copy)                      : 0 ulp vs the oracle (correctly rounded everywhere).
# Reduce Min/Max/Sum/Prod/SqNorm               : 0 ulp vs the oracle: the HIP kernel keeps the reference's order
#                                                (reduce.wgsl:68-87) and rounds x*x separately like the oracle does.
# Gemm / Gemv f32 (summation order differs from the WGSL orders: MFMA-blocked / per-lane + butterfly):
#     |gpu - f64 truth|  <= GATE_C * sqrt(K) * 2^-24 * sum_k |a||b|           (GATE_C = 2; hard bound is K * 2^-24 * ...)
#     |gpu - oracle|     <= 2 * that                                           (both sit within the gate of the truth)
#   and, at the reference's own test shapes, the reference's literal bar: abs <= 1e-3 (gemm.rs:199, gemv.rs:194).
# f16 Gemm (extension, no reference kernel)    : f16 inputs, f32 accumulate, one RNE rounding to f16:
#     |gpu - f64 truth|  <= f32 gate + 2^-11 * |truth|   (half an f16 ulp of the result; 2^-24 floor for subnormals)
GATE_C = 2.0
REF_ABS_EPS = 1.0e-3


def f32_gate(k: int, sabs) -> np.ndarray:
    return GATE_C * np.sqrt(max(int(k), 1)) * 2.0 ** -24 * np.asarray(sabs, np.float64) + 1e-37


def assert_close_f64(got, truth, k, sabs, what=""):
    got = np.asarray(got, np.float64).ravel()
    truth = np.asarray(truth, np.float64).ravel()
    tol = f32_gate(k, np.asarray(sabs).ravel())
    err = np.abs(got - truth)
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()} of {bad.size} elements exceed {GATE_C}*sqrt({k})*2^-24*sum|a||b|; "
                           f"worst err/tol = {(err / tol).max():.3g}, max abs err = {err.max():.3g}")
    return float((err / tol).max())


def assert_close_oracle(got, oracle, k, sabs, what=""):
    got = np.asarray(got, np.float64).ravel()
    oracle = np.asarray(oracle, np.float64).ravel()
    tol = 2.0 * f32_gate(k, np.asarray(sabs).ravel())
    err = np.abs(got - oracle)
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} elements differ from the oracle by more than 2x the gate; worst {(err / tol).max():.3g}"


def max_ulp(a, b) -> int:
    a = np.asarray(a, np.float32).ravel()
    b = np.asarray(b, np.float32).ravel()
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return int(np.abs(ia - ib).max()) if a.size else 0


def assert_bits_equal(got, expected, what=""):
    got = np.ascontiguousarray(got)
    expected = np.ascontiguousarray(expected)
    assert got.dtype == expected.dtype and got.shape == expected.shape, (what, got.dtype, expected.dtype, got.shape, expected.shape)
    if got.tobytes() != expected.tobytes():
        g32, e32 = got.view(np.uint32 if got.dtype.itemsize == 4 else np.uint16), expected.view(np.uint32 if got.dtype.itemsize == 4 else np.uint16)
        idx = np.flatnonzero(g32.ravel() != e32.ravel())
        raise AssertionError(f"{what}: {idx.size} of {got.size} elements differ bitwise; first at {idx[0]}: "
                             f"got {got.ravel()[idx[0]]!r}, expected {expected.ravel()[idx[0]]!r}")


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def relative_eq(a, b, epsilon, max_relative=np.finfo(np.float32).eps):
    """approx::assert_relative_eq! semantics: |a-b| <= epsilon or |a-b| <= max_relative * max(|a|,|b|)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = np.abs(a - b)
    return bool(np.all((d <= epsilon) | (d <= max_relative * np.maximum(np.abs(a), np.abs(b)))))
