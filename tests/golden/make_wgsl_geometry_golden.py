"""Generates tests/golden/wgsl_exec_geometry.npz by EXECUTING THE REFERENCE'S OWN GEOMETRY SHADER TEXT with oracle/wgsl_exec.py.

    python tests/golden/make_wgsl_geometry_golden.py [/root/reference]

Needs the reference checkout (reads crates/wgebra/src/geometry/{inv,quat,rot2,sim2,sim3,cholesky,lu}.wgsl and utils/trig.wgsl where they lie;
nothing of them is copied into the repo); the fixture holds data only: seeded inputs and the values the shader functions returned, in the item
layouts of wg_geometry_apply (wgmath_amd/csrc/geometry_items.hpp). tests/test_geometry.py compares the host build of include/wgebra_geometry.hpp
(CPU suite) and the HIP kernels (-m gpu) with it: 0 ulp for everything built from + - * / sqrt, <= 2 ulp where sin / cos enter.

Reading of the WGSL the executor fixes (WGSL leaves it to the implementation): expressions left to right, every product and sum rounded to
f32 (no FMA contraction); dot(a, b) = a.x b.x + a.y b.y + ..; cross(a, b) = (a.y b.z - b.y a.z, a.z b.x - b.z a.x, a.x b.y - b.x a.y);
length = sqrt(dot); sqrt correctly rounded; sin / cos = the float64 value rounded to f32.
Macros (naga_oil shader defs, cholesky.rs / lu.rs): DIM / NROWS / NCOLS = 2u, 3u, 4u; MAT = matNxN<f32>; PERM = vecN<u32>.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import wgsl_exec as wx  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
GEO = os.path.join(REF, "crates", "wgebra", "src", "geometry")
f32 = np.float32
COUNT = 64


def src(name, d=GEO):
    return open(os.path.join(d, name)).read()


def upm(rng, *shape):
    return (rng.random(shape, dtype=np.float32) * f32(2) - f32(1)).astype(np.float32)


trig = wx.Module(src("trig.wgsl", os.path.join(REF, "crates", "wgebra", "src", "utils")))
inv = wx.Module(src("inv.wgsl"))
rot2 = wx.Module(src("rot2.wgsl"), imports={"Trig": trig})
quat = wx.Module(src("quat.wgsl"))
sim2 = wx.Module(src("sim2.wgsl"), imports={"Rot": rot2})
sim3 = wx.Module(src("sim3.wgsl"), imports={"Rot": quat})
chol = {n: wx.Module(src("cholesky.wgsl"), subst={"DIM": f"{n}u", "MAT": f"mat{n}x{n}<f32>", "IMPORT_PATH": "x"}) for n in (2, 3, 4)}
lu = {n: wx.Module(src("lu.wgsl"), subst={"NROWS": f"{n}u", "NCOLS": f"{n}u", "PERM": f"vec{n}<u32>", "MAT": f"mat{n}x{n}<f32>", "IMPORT_PATH": "x"})
      for n in (2, 3, 4)}
S = wx.Struct
out = {}
rng = np.random.default_rng(20260405)


def flat(*vals):
    return np.concatenate([np.asarray(v, np.float32).reshape(-1) for v in vals])


# ---- matrices: item = N x N column-major floats (m[c][r] at c * N + r), exactly the (N, N) "array of columns" the executor holds
for n in (2, 3, 4):
    m = upm(rng, COUNT, n, n)
    m[:8] += np.eye(n, dtype=np.float32) * f32(2)              # some well-conditioned ones, the rest as they come
    out[f"inv{n}_in"] = m.reshape(COUNT, -1)
    out[f"inv{n}_out"] = np.stack([inv.call_fn(f"inv{n}", m[i]).reshape(-1) for i in range(COUNT)])
    b = upm(rng, COUNT, n, n)
    spd = np.stack([(b[i].astype(np.float64) @ b[i].astype(np.float64).T + np.eye(n)).astype(np.float32) for i in range(COUNT)])
    out[f"cholesky{n}_in"] = spd.reshape(COUNT, -1)
    out[f"cholesky{n}_out"] = np.stack([chol[n].call_fn("cholesky", spd[i]).reshape(-1) for i in range(COUNT)])
    a = upm(rng, COUNT, n, n)
    a[:4, :, 0] = 0                                             # a zero first row: the pivot search must move on (and a `continue` for column 0 of item 0)
    a[0, 0, :] = 0
    res = []
    for i in range(COUNT):
        r = lu[n].call_fn("lu", a[i])
        res.append(flat(r.lu, np.array(r.p.ia, np.float32), np.array(r.p.ib, np.float32), [f32(r.p.len)]))
    out[f"lu{n}_in"] = a.reshape(COUNT, -1)
    out[f"lu{n}_out"] = np.stack(res)

# ---- Quat: arbitrary (NOT unit) coordinates -- a unit quaternion times 1 +- a few per cent, and some far from unit
qa, qb, v3 = upm(rng, COUNT, 4), upm(rng, COUNT, 4), upm(rng, COUNT, 3) * f32(3)
qa[: COUNT // 2] = (qa[: COUNT // 2] / np.linalg.norm(qa[: COUNT // 2], axis=1, keepdims=True) * (1 + 0.05 * upm(rng, COUNT // 2, 1))).astype(np.float32)
rows = []
for i in range(COUNT):
    A, B = S(["coords"], [qa[i]]), S(["coords"], [qb[i]])
    rows.append(flat(quat.call_fn("mul", A, B).coords, quat.call_fn("mulVec", A, v3[i]), quat.call_fn("invMulVec", A, v3[i]), quat.call_fn("toMatrix", A),
                     quat.call_fn("renormalizeFast", A).coords, quat.call_fn("inv", A).coords))
out["quat_raw_in"] = np.concatenate([qa, qb, v3], axis=1)
out["quat_raw_out"] = np.stack(rows)

# ---- Rot2: arbitrary (cos, sin) pairs
ra, rb, v2 = upm(rng, COUNT, 2), upm(rng, COUNT, 2), upm(rng, COUNT, 2) * f32(3)
rows = []
for i in range(COUNT):
    A, B = S(["cos_sin"], [ra[i]]), S(["cos_sin"], [rb[i]])
    rows.append(flat(rot2.call_fn("mul", A, B).cos_sin, rot2.call_fn("mulVec", A, v2[i]), rot2.call_fn("invMulVec", A, v2[i]), rot2.call_fn("toMatrix", A),
                     rot2.call_fn("inv", A).cos_sin))
out["rot2_raw_in"] = np.concatenate([ra, rb, v2], axis=1)
out["rot2_raw_out"] = np.stack(rows)

# ---- Sim2: (cos, sin, tx, ty, scale) x 2 + a point
sa = np.concatenate([upm(rng, COUNT, 2), upm(rng, COUNT, 2) * f32(5), f32(0.5) + rng.random((COUNT, 1), dtype=np.float32) * f32(2)], axis=1)
sb = np.concatenate([upm(rng, COUNT, 2), upm(rng, COUNT, 2) * f32(5), f32(0.5) + rng.random((COUNT, 1), dtype=np.float32) * f32(2)], axis=1)
pt2 = upm(rng, COUNT, 2) * f32(4)
F2 = ["rotation", "translation", "scale"]


def s2(x):
    return S(F2, [S(["cos_sin"], [x[0:2].copy()]), x[2:4].copy(), f32(x[4])])


def s2flat(s):
    return flat(s.rotation.cos_sin, s.translation, [s.scale])


rows = []
for i in range(COUNT):
    A, B = s2(sa[i]), s2(sb[i])
    rows.append(flat(s2flat(sim2.call_fn("mul", A, B)), s2flat(sim2.call_fn("inv", A)), sim2.call_fn("mulPt", A, pt2[i]), sim2.call_fn("invMulPt", A, pt2[i]),
                     sim2.call_fn("mulVec", A, pt2[i]), sim2.call_fn("invMulVec", A, pt2[i])))
out["sim2_raw_in"] = np.concatenate([sa, sb, pt2], axis=1)
out["sim2_raw_out"] = np.stack(rows)

# ---- Sim3: (q[4], t[3], scale) x 2 + a point; the reference packs (t, scale) into one vec4 `translation_scale`
ta = np.concatenate([upm(rng, COUNT, 4), upm(rng, COUNT, 3) * f32(5), f32(0.5) + rng.random((COUNT, 1), dtype=np.float32) * f32(2)], axis=1)
tb = np.concatenate([upm(rng, COUNT, 4), upm(rng, COUNT, 3) * f32(5), f32(0.5) + rng.random((COUNT, 1), dtype=np.float32) * f32(2)], axis=1)
pt3 = upm(rng, COUNT, 3) * f32(4)
F3 = ["rotation", "translation_scale"]


def s3(x):
    return S(F3, [S(["coords"], [x[0:4].copy()]), x[4:8].copy()])


def s3flat(s):
    return flat(s.rotation.coords, s.translation_scale)


rows = []
for i in range(COUNT):
    A, B = s3(ta[i]), s3(tb[i])
    rows.append(flat(s3flat(sim3.call_fn("mul", A, B)), s3flat(sim3.call_fn("inv", A)), sim3.call_fn("mulPt", A, pt3[i]), sim3.call_fn("invMulPt", A, pt3[i]),
                     sim3.call_fn("mulVec", A, pt3[i]), sim3.call_fn("invMulVec", A, pt3[i])))
out["sim3_raw_in"] = np.concatenate([ta, tb, pt3], axis=1)
out["sim3_raw_out"] = np.stack(rows)

# ---- sin / cos enter: fromScaledAxis (quat.wgsl:16-28), fromAngle (rot2.wgsl:20-22); the zero axis is the identity
ax = upm(rng, COUNT, 3) * f32(2.5)
ax[0] = 0
ang = upm(rng, COUNT, 1) * f32(6)
out["from_in"] = np.concatenate([ax, ang], axis=1)
out["from_out"] = np.stack([flat(quat.call_fn("fromScaledAxis", ax[i]).coords, rot2.call_fn("fromAngle", f32(ang[i, 0])).cos_sin) for i in range(COUNT)])

path = os.path.join(HERE, "wgsl_exec_geometry.npz")
np.savez_compressed(path, **{k: np.ascontiguousarray(v, np.float32) for k, v in out.items()})
print(f"wgsl_exec_geometry.npz  {os.path.getsize(path) / 1024:.1f} KiB;", {k: v.shape for k, v in out.items() if k.endswith('_out')})
