"""Generates tests/golden/wgsl_exec_decomp.npz by EXECUTING THE REFERENCE'S OWN SHADER TEXT with oracle/wgsl_exec.py: the decompositions and the
two utils modules -- crates/wgebra/src/geometry/{qr2,qr3,qr4,eig2,eig3,eig4,svd2,svd3,rot2}.wgsl and crates/wgebra/src/utils/{trig,min_max}.wgsl.

    python tests/golden/make_wgsl_decomp_golden.py [/root/reference]

Needs the reference checkout (the .wgsl files are read where they lie; nothing of them is copied into the repo). The fixture holds data only: seeded
inputs and the values the shader functions returned, in the item layouts of wg_geometry_apply (wgmath_amd/csrc/geometry_items.hpp). tests/test_geometry.py
compares the host build of include/wgebra_geometry.hpp (CPU suite) and the HIP kernels (-m gpu) with it bit for bit.

Reading of the WGSL the executor fixes (the header fixes the same one): as make_wgsl_geometry_golden.py, plus: `fma` = one rounding; atan / exp (like
sin / cos) = the float64 value rounded to f32; sign(+-0) = +0; matN * matN element (i, j) = a[0][i] b[j][0] + a[1][i] b[j][1] + .. left to right.

Inputs per size: the reference's own test distribution (nalgebra `new_random`, U[0, 1): qr3.rs:100, eig3.rs:108, svd3.rs:100), U[-1, 1), and the edge
items the reference wrote code for: identity, zero, diagonal (eig2.wgsl:20 `c == 0`; delimit_subproblem's immediate (0, 0)), a zero first column
(`factor == 0` in the Householder step), rank one and rank two (svd3's epsilon branches), block-diagonal (decoupling), negated identity, permutations.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import wgsl_exec as wx  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
SRC = os.path.join(REF, "crates", "wgebra", "src")
f32 = np.float32
S = wx.Struct


def src(rel):
    return open(os.path.join(SRC, rel)).read()


trig = wx.Module(src("utils/trig.wgsl"))
min_max = wx.Module(src("utils/min_max.wgsl"))
rot2 = wx.Module(src("geometry/rot2.wgsl"), imports={"Trig": trig})
quat = wx.Module(src("geometry/quat.wgsl"))
eig = {2: wx.Module(src("geometry/eig2.wgsl"))}
for n in (3, 4):
    eig[n] = wx.Module(src(f"geometry/eig{n}.wgsl"), imports={"Rot": rot2, "Eig2": eig[2], "MinMax": min_max})
qr = {n: wx.Module(src(f"geometry/qr{n}.wgsl")) for n in (2, 3, 4)}
svd = {2: wx.Module(src("geometry/svd2.wgsl"), imports={"Trig": trig}), 3: wx.Module(src("geometry/svd3.wgsl"), imports={"Quat": quat})}

rng = np.random.default_rng(20261002)
out = {}


def u01(*shape):
    return rng.random(shape, dtype=np.float32)


def upm(*shape):
    return (rng.random(shape, dtype=np.float32) * f32(2) - f32(1)).astype(np.float32)


def flat(*vals):
    return np.concatenate([np.asarray(v, np.float32).reshape(-1) for v in vals])


def edge_matrices(n):
    """Matrices as (n, n) arrays of COLUMNS (m[c][r]), the executor's and the item layout's order."""
    eye = np.eye(n, dtype=np.float32)
    e = [eye.copy(), np.zeros((n, n), np.float32), -eye, np.diag(np.arange(1, n + 1).astype(np.float32)), np.diag(np.arange(n, 0, -1).astype(np.float32) * f32(-0.5))]
    m = upm(n, n); m[0, :] = 0; e.append(m)                                  # a zero first column
    m = upm(n, n); m[:, 0] = 0; e.append(m)                                  # a zero first row
    v, w = upm(n), upm(n); e.append(np.outer(v, w).astype(np.float32))       # rank one
    e.append(np.outer(v, v).astype(np.float32))                              # rank one, symmetric
    if n > 2:
        e.append((np.outer(v, w) + np.outer(upm(n), upm(n))).astype(np.float32))  # rank two
    e.append(np.triu(upm(n, n)).astype(np.float32))                          # columns-array upper triangle = a lower-triangular matrix
    e.append(np.tril(upm(n, n)).astype(np.float32))                          # an upper-triangular matrix (QR: nothing to annihilate below the diagonal)
    e.append(eye[::-1].copy())                                               # the exchange permutation
    e.append(np.roll(eye, 1, axis=0).copy())                                 # a cyclic permutation
    m = upm(n, n); m[0, 1:] = 0; m[1:, 0] = 0; e.append(m)                   # block-diagonal 1 + (n - 1)
    e.append(np.full((n, n), f32(0.25)))                                     # all entries equal (rank one, repeated eigenvalue 0)
    e.append((eye * f32(3) + np.full((n, n), f32(1e-4))).astype(np.float32))  # nearly a multiple of the identity
    e.append((upm(n, n) * f32(1e-12)).astype(np.float32))                    # tiny entries (eig: the scaling by the largest |entry|)
    e.append((upm(n, n) * f32(1e12)).astype(np.float32))                     # huge entries
    return np.stack(e)


for n in (2, 3, 4):
    mats = np.concatenate([u01(48, n, n), upm(32, n, n), edge_matrices(n)])
    count = len(mats)
    # ---- QR (qr2.wgsl:15-107, qr3.wgsl:15-109, qr4.wgsl:15-111)
    res = [qr[n].call_fn("qr", mats[i]) for i in range(count)]
    out[f"qr{n}_in"] = mats.reshape(count, -1)
    out[f"qr{n}_out"] = np.stack([flat(r.q, r.r) for r in res])
    # ---- symmetric eigen (eig2.wgsl:15-42, eig3.wgsl:24-160, eig4.wgsl:24-162): symmetrised inputs (exactly symmetric: (m + m^T) / 2 in f32)
    sym = ((mats + mats.transpose(0, 2, 1)) * f32(0.5)).astype(np.float32)
    res = [eig[n].call_fn("symmetric_eigen", sym[i]) for i in range(count)]
    out[f"sym_eigen{n}_in"] = sym.reshape(count, -1)
    out[f"sym_eigen{n}_out"] = np.stack([flat(r.eigenvectors, r.eigenvalues) for r in res])
    if n == 2:
        out["eigvals2_in"] = sym.reshape(count, -1)
        out["eigvals2_out"] = np.stack([flat(eig[2].call_fn("eigenvalues", sym[i])) for i in range(count)])
    # ---- SVD + recompose (svd2.wgsl:12-46, svd3.wgsl:296-312)
    if n < 4:
        res = [svd[n].call_fn("svd", mats[i]) for i in range(count)]
        out[f"svd{n}_in"] = mats.reshape(count, -1)
        out[f"svd{n}_out"] = np.stack([flat(r.U, r.S, r.Vt) for r in res])
        out[f"svd_recompose{n}_in"] = out[f"svd{n}_out"].copy()
        out[f"svd_recompose{n}_out"] = np.stack([flat(svd[n].call_fn("recompose", r)) for r in res])

# ---- utils: trig (trig.wgsl:12-38) + min_max (min_max.wgsl:4-51). Item: y, x, t, m[16]
COUNT = 96
yx = upm(COUNT, 2) * f32(3)
edge_yx = np.array([[0, -1], [0, -2.5], [1, 0], [-1, 0], [0, 0], [0, 1], [1, -1], [-1, -1], [1, 1], [-1, 1], [-0.0, -1], [1e-30, -1], [-1e-30, -1], [3, -1e-30],
                    [0.0, -0.0]], np.float32)
yx[:len(edge_yx)] = edge_yx
t = upm(COUNT, 1) * f32(4)
t[:10, 0] = np.array([0, -0.0, 50, -50, 100, -100, 1e-8, -1e-8, 9.5, -9.5], np.float32)
m16 = upm(COUNT, 16) * f32(10)
m16[0] = 0
m16[1] = -np.abs(m16[1])                      # all negative: max != amax
m16[2, :] = f32(-7.5)
rows = []
for i in range(COUNT):
    m = m16[i]
    rows.append(flat([trig.call_fn("stable_atan2", yx[i, 0], yx[i, 1]), trig.call_fn("stable_tanh", t[i, 0]),
                      min_max.call_fn("max2", m[:2]), min_max.call_fn("amax2x2", m[:4].reshape(2, 2)), min_max.call_fn("max2x2", m[:4].reshape(2, 2)),
                      min_max.call_fn("max3", m[:3]), min_max.call_fn("amax3x3", m[:9].reshape(3, 3)), min_max.call_fn("max3x3", m[:9].reshape(3, 3)),
                      min_max.call_fn("max4", m[:4]), min_max.call_fn("amax4x4", m.reshape(4, 4)), min_max.call_fn("max4x4", m.reshape(4, 4))]))
out["utils_in"] = np.concatenate([yx, t, m16], axis=1)
out["utils_out"] = np.stack(rows)

# ---- the Rot2 functions the eigen-solvers use (rot2.wgsl:15-36, :51-53, :78-95). Item: rot (cos, sin), v, m3[9], m4[16], i3, i4
rot = upm(COUNT, 2)
rot[:6] = np.array([[-1, 0], [1, 0], [0, 1], [0, -1], [0, 0], [-0.5, 0]], np.float32)   # angle(): 0 (not pi) for (-1, 0); 0 for cos == 0
v = upm(COUNT, 2) * f32(2)
v[:5] = np.array([[1, 0], [0, 0], [0, 1], [-2, 1e-20], [0, -3]], np.float32)            # cancel_y: the zero Rot2 for v.y == 0; sign(0) / |v| for v.x == 0
m3, m4 = upm(COUNT, 9), upm(COUNT, 16)
i3, i4 = rng.integers(0, 2, COUNT), rng.integers(0, 3, COUNT)
rows = []
for i in range(COUNT):
    R = S(["cos_sin"], [rot[i].copy()])
    cy = rot2.call_fn("cancel_y", v[i])
    p3, p4 = wx.Ref(m3[i].reshape(3, 3).copy()), wx.Ref(m4[i].reshape(4, 4).copy())
    rot2.call_fn("rotate_rows3", R, p3, int(i3[i]))
    rot2.call_fn("rotate_rows4", R, p4, int(i4[i]))
    rows.append(flat([rot2.call_fn("angle", R)], cy.cos_sin, [f32(bool(rot2.call_fn("is_valid", cy)))], p3.v, p4.v))
out["rot2_ext_in"] = np.concatenate([rot, v, m3, m4, i3[:, None].astype(np.float32), i4[:, None].astype(np.float32)], axis=1)
out["rot2_ext_out"] = np.stack(rows)

path = os.path.join(HERE, "wgsl_exec_decomp.npz")
np.savez_compressed(path, **{k: np.ascontiguousarray(v, np.float32) for k, v in out.items()})
print(f"wgsl_exec_decomp.npz  {os.path.getsize(path) / 1024:.1f} KiB;", {k: v.shape for k, v in out.items() if k.endswith('_out')})
for k, v in out.items():
    if k.endswith("_out") and not np.isfinite(v).all():
        bad = np.argwhere(~np.isfinite(v).all(axis=1)).reshape(-1)
        print(f"  {k}: non-finite values in items {bad.tolist()}")
