"""A small executor for the WGSL subset the reference's dense linear-algebra shaders are written in.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): nothing under wgmath_amd/ or include/ may import it.

Purpose: pin the CPU restatement (oracle/wgsl_oracle.{c,py}) against the reference's OWN shader text. The reference is
Rust + WGSL executed by wgpu/naga; neither exists in this environment, so the shaders cannot be run "for real". What can be
done is to take the .wgsl files as they are (crates/wgebra/src/linalg/{shape,gemm,gemv,reduce,op_assign}.wgsl), translate them
mechanically into Python generators (one generator per invocation, `workgroupBarrier()` = yield, the invocations of a workgroup
advanced in lock-step) and run them on f32 NumPy values. tests/golden/make_wgsl_golden.py does that HERE (where /root/reference
is mounted) and commits inputs + outputs as fixtures; tests/test_oracle.py checks both restatements against them bit for bit.
A transcription slip in the restatement (a wrong stride, an off-by-one in a tree, a swapped index) shows up as a mismatch.

What this executor fixes that WGSL leaves to the implementation (and the restatement fixes the same way):
  * `mat4x4 * mat4x4`, `mat4x4 * vec4`: column-major, the four products of an output element are added left to right,
    every product and every sum rounded to f32 (no FMA contraction);
  * u32 arithmetic wraps modulo 2^32; f32 arithmetic is IEEE round-to-nearest-even (NumPy float32).

Supported (all that the five files use): `#import .. as ..`, `#define_import_path`, `#ifdef/#else/#endif`, struct declarations,
module-scope `const`, `var<uniform>`, `var<storage,...>`, `var<workgroup>`, functions, `@compute` entry points with
`@builtin(global_invocation_id | local_invocation_id | workgroup_id)`, `let`/`var`, assignment and compound assignment, `x++`,
`if`/`else`, `for`, `return`, calls, `vec4`, `mat4x4`, `mat4x4f`, `transpose`, `min`, `max`, member access, indexing, and the
function redirection the Rust side performs through naga_oil's `Redirector` (reduce.rs:74-91, op_assign.rs:59-63).

Round 4: the geometry shader library (crates/wgebra/src/geometry/{inv,quat,rot2,sim2,sim3,cholesky,lu}.wgsl) runs through the same
translator -- called function by function (`call_fn`), no entry point: vec2/3/4 and mat2x2/3x3/4x4 values with mixed-argument constructors and
swizzles, `dot` / `cross` / `length` as their defining formulas (left to right, every product and sum rounded to f32), `sqrt` correctly rounded,
`sin` / `cos` rounded from float64, zero-value constructors (`Rot2()`), `var x: T;`, element writes `m[i][j] = ..`, `ptr<function, T>`
parameters (`&m`, `(*m)`), `continue`, struct member writes, and the textual macro substitution (DIM, MAT, NROWS, ...) the Rust side does
through naga_oil shader defs (cholesky.rs, lu.rs). tests/golden/make_wgsl_geometry_golden.py writes the fixtures.

Round 5: the rest of the geometry library and the two utils modules (qr2/3/4, eig2/3/4, svd2/3, utils/trig.wgsl, utils/min_max.wgsl): `while`,
`break` (and `continue`) in both loop forms, function-scope `const`, block-scoped shadowing (`let m = ..` inside a loop whose function already has an
`m`), `ptr<function, T>` as real references (`&x` = a cell that is written back to the local after the statement, `*p` reads / writes the cell: what
`condSwap(c, &rho1, &rho2)` on two f32 locals needs), `select` (scalar and vector conditions), `fma` (ONE rounding: the exact product plus the addend
rounded once, computed through a round-to-odd float64 sum), `bitcast<i32>` / `bitcast<f32>`, `>>` on i32, hexadecimal literals, `array<f32, N>(..)`,
integer vectors (`vec2<u32>`), matN x matN / matN x vecN products for N = 2, 3, 4 (left to right, every product and sum rounded), `atan` / `exp` as the
float64 value rounded to f32 (like `sin` / `cos`), `sign(+-0) = +0`, module-scope constants of imported modules (`Trig::PI` inside `stable_atan2`).
"""
from __future__ import annotations

import re

import numpy as np

M32 = 0xFFFFFFFF
f32 = np.float32


# ------------------------------------------------------------------------------------------------------------------
# runtime
# ------------------------------------------------------------------------------------------------------------------
def _is_int(x):
    return isinstance(x, int) and not isinstance(x, bool)


def _num(x):
    """Abstract numeric literals meet f32 values: promote Python numbers to float32."""
    return x if isinstance(x, (np.ndarray, np.floating)) else f32(x)


def _matmul(a, b):
    """a: matNxN as an (N, N) array of COLUMNS (a[k] = column k); b: matNxN (columns) or vecN."""
    if b.ndim == 1:
        r = a[0] * b[0]
        for k in range(1, a.shape[0]):
            r = r + a[k] * b[k]
        return r
    return np.stack([_matmul(a, b[j]) for j in range(b.shape[0])])


def op_add(a, b):
    if _is_int(a) and _is_int(b):
        return (a + b) & M32
    return _num(a) + _num(b)


def op_sub(a, b):
    if _is_int(a) and _is_int(b):
        return (a - b) & M32
    return _num(a) - _num(b)


def op_mul(a, b):
    if _is_int(a) and _is_int(b):
        return (a * b) & M32
    a, b = _num(a), _num(b)
    if isinstance(a, np.ndarray) and a.ndim == 2 and isinstance(b, np.ndarray):
        return _matmul(a, b)
    return a * b


def op_div(a, b):
    if _is_int(a) and _is_int(b):
        return a // b
    return _num(a) / _num(b)


def fn_transpose(m):
    return np.ascontiguousarray(m.T)


def fn_min(a, b):
    if _is_int(a) and _is_int(b):
        return min(a, b)
    return np.minimum(_num(a), _num(b))


def fn_max(a, b):
    if _is_int(a) and _is_int(b):
        return max(a, b)
    return np.maximum(_num(a), _num(b))


def fn_vec4(*args):
    if len(args) == 1:
        return np.full(4, _num(args[0]), f32)
    return np.array([_num(a) for a in args], f32)


def fn_mat4x4(*cols):
    if not cols:
        return np.zeros((4, 4), f32)
    return np.stack([np.asarray(c, f32) for c in cols])


def load(arr, i):
    v = arr[i]
    return v.copy() if isinstance(v, np.ndarray) else v


def op_neg(a):
    if _is_int(a):
        return (-a) & M32
    return np.negative(_num(a))  # (0 - x would turn -0.0 into +0.0)


def fn_vec(n, *args):
    """vecN(...): a splat, N scalars, or any mix of scalars and shorter vectors (vec4(axis * hs, hc))."""
    flat = []
    for a in args:
        if isinstance(a, np.ndarray):
            flat.extend(a.tolist() if a.dtype != np.float32 else list(a))
        else:
            flat.append(a)
    if not flat:
        return np.zeros(n, f32)
    if len(flat) == 1:
        flat = flat * n
    assert len(flat) == n, (n, args)
    if all(isinstance(x, (bool, np.bool_)) for x in flat):  # vec3(c) with a bool: the condition of a component-wise select
        return np.array(flat, bool)
    if all(_is_int(x) for x in flat):                        # vec2(0u, 0u), vec2(new_start, n): a vecN<u32>
        return IVec(flat)
    return np.array([_num(x) for x in flat], f32)


def fn_mat(n, *cols):
    """matNxN(col0, col1, ...) (columns) or matNxN() = zeros. Stored as an (N, N) array of COLUMNS: m[c][r]."""
    if not cols:
        return np.zeros((n, n), f32)
    if len(cols) == n * n:
        return np.array([_num(x) for x in cols], f32).reshape(n, n)
    assert len(cols) == n, (n, cols)
    return np.stack([np.asarray(c, f32) for c in cols])


_SW = {"x": 0, "y": 1, "z": 2, "w": 3}


class IVec(list):
    """vecN<u32> / vecN<i32>: Python ints (u32 arithmetic stays exact integer arithmetic), indexable and swizzle-readable."""


class Ref:
    """A `ptr<function, T>`: `&x` makes one around the local's value, `*p` is `.v`; the caller copies `.v` back into the local after the statement."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v


def member(obj, name):
    if isinstance(obj, IVec):
        return obj[_SW[name]]
    if isinstance(obj, np.ndarray):
        idx = [_SW[ch] for ch in name]
        return obj[idx[0]] if len(idx) == 1 else obj[idx].copy()
    return getattr(obj, name)


def setmember(obj, name, value):
    if isinstance(obj, np.ndarray):
        obj[_SW[name]] = value
    else:
        setattr(obj, name, value)


def copyval(v):
    """WGSL values are copied on `let` / `var` (arrays and structs are mutable objects here)."""
    if isinstance(v, np.ndarray):
        return v.copy()
    if isinstance(v, Struct):
        return Struct(v._fields, [copyval(getattr(v, f)) for f in v._fields])
    if isinstance(v, IVec):
        return IVec(v)
    if isinstance(v, list):
        return list(v)
    return v


def fn_dot(a, b):
    r = a[0] * b[0]
    for k in range(1, len(a)):
        r = r + a[k] * b[k]
    return r


def fn_cross(a, b):
    return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], f32)


def fn_sqrt(x):
    return np.sqrt(_num(x))  # IEEE: correctly rounded


def fn_length(v):
    return np.sqrt(fn_dot(v, v))


def fn_sin(x):
    return f32(np.sin(np.float64(_num(x))))


def fn_cos(x):
    return f32(np.cos(np.float64(_num(x))))


def fn_atan(x):
    return f32(np.arctan(np.float64(_num(x))))


def fn_exp(x):
    with np.errstate(over="ignore"):
        return f32(np.exp(np.float64(_num(x))))


def fn_abs(x):
    return abs(x) if _is_int(x) else np.abs(_num(x))


def fn_sign(x):
    """1 for x > 0, -1 for x < 0, +0 otherwise (WGSL leaves the sign of sign(-0) open; the header fixes it the same way)."""
    x = _num(x)
    one = f32(1)
    if isinstance(x, np.ndarray):
        return np.where(x > 0, one, np.where(x < 0, -one, f32(0))).astype(f32)
    return one if x > 0 else (-one if x < 0 else f32(0))


def fn_select(f, t, cond):
    """select(f, t, cond): t where cond, else f; component-wise for a vector condition."""
    if isinstance(cond, np.ndarray):
        return np.where(cond, t, f).astype(np.asarray(t).dtype)
    return copyval(t) if cond else copyval(f)


def fn_fma(a, b, c):
    """fma(a, b, c) with ONE rounding. The f32 x f32 product is exact in float64; the float64 sum p + c is rounded to odd (its TwoSum error says on
    which side of the float64 sum the exact value lies), and a round-to-odd float64 rounds to f32 exactly as the exact value does (53 >= 24 + 2)."""
    a, b, c = np.float64(_num(a)), np.float64(_num(b)), np.float64(_num(c))
    p = a * b
    s = p + c
    if not np.isfinite(s):
        return f32(s)
    bb = s - p
    err = (p - (s - bb)) + (c - bb)
    if err != 0.0 and (int(np.float64(s).view(np.int64)) & 1) == 0:
        s = np.nextafter(s, np.float64(np.inf) if err > 0 else np.float64(-np.inf))
    return f32(s)


def fn_bitcast_i32(x):
    v = int(np.float32(x).view(np.int32)) if not _is_int(x) else x & M32
    return v - (1 << 32) if v >= (1 << 31) else v  # signed: `i >> 1` on it is Python's arithmetic shift


def fn_bitcast_u32(x):
    return int(np.float32(x).view(np.uint32)) if not _is_int(x) else x & M32


def fn_bitcast_f32(i):
    return np.uint32(int(i) & M32).view(np.float32) if _is_int(i) else f32(i)


def fn_array(*args):
    """array<T, N>(a, b, ..): a fixed-size array of scalars (f32 here) or of whatever the elements are."""
    if args and all(not isinstance(a, (np.ndarray, Struct)) and not _is_int(a) for a in args):
        return np.array([_num(a) for a in args], f32)
    return [copyval(a) for a in args]


def op_shr(a, b):
    return a >> b


def fn_f32(x):
    return f32(x)


def fn_u32(x):
    return int(x) & M32


class Vec3:
    __slots__ = ("x", "y", "z")

    def __init__(self, x, y, z):
        self.x, self.y, self.z = int(x), int(y), int(z)


class Struct:
    def __init__(self, fields, values):
        assert len(fields) == len(values), (fields, values)
        self._fields = list(fields)
        for f, v in zip(fields, values):
            setattr(self, f, v)


# ------------------------------------------------------------------------------------------------------------------
# lexer / parser  ->  Python source
# ------------------------------------------------------------------------------------------------------------------
TOKEN = re.compile(r"\s*(?:(//[^\n]*)|(0x[0-9a-fA-F]+u?|\d+\.\d*(?:[eE][+-]?\d+)?f?|\d+[eE][+-]?\d+f?|\d+[uif]?)|([A-Za-z_][A-Za-z_0-9]*)|(::|\+\+|--|\+=|-=|\*=|/=|==|!=|<=|>=|&&|\|\||->|>>|[-+*/%<>=!(){}\[\];:,.@&|]))")


def preprocess(src: str, defs: set) -> str:
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)  # block comments
    out, stack = [], []
    for line in src.splitlines():
        s = line.strip()
        if s.startswith("#ifdef"):
            stack.append(s.split()[1] in defs)
        elif s.startswith("#ifndef"):
            stack.append(s.split()[1] not in defs)
        elif s.startswith("#else"):
            stack[-1] = not stack[-1]
        elif s.startswith("#endif"):
            stack.pop()
        elif all(stack):
            out.append(line)
    return "\n".join(out)


def tokenize(src: str):
    toks, pos = [], 0
    while pos < len(src):
        m = TOKEN.match(src, pos)
        if not m:
            if src[pos:].strip() == "":
                break
            raise SyntaxError(f"cannot tokenize at {src[pos:pos + 40]!r}")
        pos = m.end()
        if m.group(1):
            continue
        toks.append(m.group(2) or m.group(3) or m.group(4))
    return toks


class Module:
    """One WGSL file translated to Python. `imports` maps an alias (e.g. 'Shape') to another Module."""

    def __init__(self, src: str, imports: dict | None = None, defs: set | None = None, redirect: dict | None = None, subst: dict | None = None):
        self.imports = imports or {}
        self.redirect = redirect or {}
        self.structs, self.consts, self.globals_, self.functions, self.entries = {}, {}, {}, {}, {}
        self.struct_types = {}
        for k, v in (subst or {}).items():  # naga_oil shader defs with values (cholesky.rs / lu.rs): whole identifiers only
            src = re.sub(r"\b" + re.escape(k) + r"\b", v, src)
        lines = []
        for line in preprocess(src, defs or set()).splitlines():
            s = line.strip()
            if s.startswith("#import") or s.startswith("#define_import_path"):
                continue
            lines.append(line)
        self.toks = tokenize("\n".join(lines))
        self.i = 0
        self.py = []
        self.pre, self.post, self.loops, self.nflag, self.nref = [], [], [], 0, 0
        self._parse_module()
        self.ns = {"op_add": op_add, "op_sub": op_sub, "op_mul": op_mul, "op_div": op_div, "load": load, "Struct": Struct,
                   "fn_transpose": fn_transpose, "fn_min": fn_min, "fn_max": fn_max, "fn_vec4": fn_vec4, "fn_mat4x4": fn_mat4x4,
                   "f32": f32, "M32": M32, "MOD": self, "op_neg": op_neg, "fn_vec": fn_vec, "fn_mat": fn_mat, "member": member,
                   "setmember": setmember, "copyval": copyval, "fn_dot": fn_dot, "fn_cross": fn_cross, "fn_sqrt": fn_sqrt,
                   "fn_length": fn_length, "fn_sin": fn_sin, "fn_cos": fn_cos, "fn_abs": fn_abs, "fn_sign": fn_sign, "fn_f32": fn_f32,
                   "fn_u32": fn_u32, "zero_of": self.zero_of, "fn_atan": fn_atan, "fn_exp": fn_exp, "fn_select": fn_select, "fn_fma": fn_fma,
                   "fn_bitcast_i32": fn_bitcast_i32, "fn_bitcast_u32": fn_bitcast_u32, "fn_bitcast_f32": fn_bitcast_f32, "fn_array": fn_array,
                   "op_shr": op_shr, "Ref": Ref, "IVec": IVec}
        for alias, mod in self.imports.items():
            self.ns["IMP_" + alias] = mod
        exec("\n".join(self.py), self.ns)
        self._constG = None

    def zero_of(self, ty: str):
        """The zero value of a WGSL type (`var x: T;`, `T()`)."""
        ty = ty.replace(" ", "")
        m = re.fullmatch(r"vec(\d)(?:<(\w+)>|f|u|i)?", ty)
        if m:
            n, el = int(m.group(1)), (m.group(2) or ("f32" if ty.endswith("f") or ty == "vec" + m.group(1) else "u32"))
            return np.zeros(n, f32) if el == "f32" else [0] * n
        m = re.fullmatch(r"mat(\d)x(\d)(?:<f32>|f)?", ty)
        if m:
            return np.zeros((int(m.group(1)), int(m.group(2))), f32)
        if ty in ("f32",):
            return f32(0)
        if ty in ("u32", "i32"):
            return 0
        if ty == "bool":
            return False
        if "::" in ty:
            alias, name = ty.split("::")
            return self.imports[alias].zero_of(name)
        if ty in self.structs:
            return Struct(self.structs[ty], [self.zero_of(t) for t in self.struct_types[ty]])
        raise TypeError(f"no zero value for type {ty!r}")

    @property
    def G(self):
        """Module-scope constants as the `G` object of functions run outside an entry point (call_fn, and calls from an importing module)."""
        if self._constG is None:
            class G:
                pass
            g = G()
            for k, v in self.consts.items():
                setattr(g, k, eval(v, self.ns, {"G": g}))
            self._constG = g
        return self._constG

    def call_fn(self, name: str, *args):
        """Runs function `name` (not an entry point) to completion and returns its value. Arguments are copied (value semantics); an argument of a
        `ptr<function, T>` parameter is passed as a wgsl_exec.Ref (read `.v` afterwards)."""
        gen = self.ns["F_" + name](self.G, *[a if isinstance(a, Ref) else copyval(a) for a in args])
        try:
            while True:
                next(gen)
        except StopIteration as e:
            return e.value

    # -- token helpers
    def peek(self, k=0):
        return self.toks[self.i + k] if self.i + k < len(self.toks) else None

    def next(self):
        t = self.toks[self.i]
        self.i += 1
        return t

    def expect(self, t):
        got = self.next()
        if got != t:
            raise SyntaxError(f"expected {t!r}, got {got!r} near {' '.join(self.toks[max(0, self.i - 8):self.i + 4])}")

    def accept(self, t):
        if self.peek() == t:
            self.i += 1
            return True
        return False

    def skip_attrs(self):
        attrs = []
        while self.peek() == "@":
            self.next()
            name = self.next()
            args = []
            if self.accept("("):
                while not self.accept(")"):
                    args.append(self.next())
            attrs.append((name, [a for a in args if a != ","]))
        return attrs

    def parse_type(self):
        """Returns the type as a flat string such as 'array<vec4<f32>,64>' or 'Shape::Shape'."""
        t = self.next()
        while self.peek() == "::":
            t += self.next() + self.next()
        if self.accept("<"):
            t += "<"
            depth = 1
            while depth:
                x = self.next()
                depth += (x == "<") - (x == ">") - 2 * (x == ">>")
                t += x
        return t

    # -- module level
    def _parse_module(self):
        while self.peek() is not None:
            attrs = self.skip_attrs()
            t = self.peek()
            if t == ";":
                self.next()
            elif t == "struct":
                self.next()
                name = self.next()
                self.expect("{")
                fields, ftypes = [], []
                while not self.accept("}"):
                    self.skip_attrs()
                    fields.append(self.next())
                    self.expect(":")
                    ftypes.append(self.parse_type())
                    self.accept(",")
                self.structs[name] = fields
                self.struct_types[name] = ftypes
            elif t == "const":
                self.next()
                name = self.next()
                if self.accept(":"):
                    self.parse_type()
                self.expect("=")
                self.consts[name] = self.parse_expr(Scope(self, None))
                self.expect(";")
            elif t == "var":
                self.next()
                space = ""
                if self.accept("<"):
                    while not self.accept(">"):
                        space += self.next()
                name = self.next()
                self.expect(":")
                ty = self.parse_type()
                self.expect(";")
                self.globals_[name] = (space, ty)
            elif t == "fn":
                self._parse_fn(attrs)
            else:
                raise SyntaxError(f"unexpected {t!r} at module scope")

    def _parse_fn(self, attrs):
        self.expect("fn")
        name = self.next()
        self.expect("(")
        params = []
        while not self.accept(")"):
            pattrs = self.skip_attrs()
            pname = self.next()
            self.expect(":")
            self.parse_type()
            self.accept(",")
            params.append((pname, pattrs))
        if self.accept("->"):
            self.skip_attrs()
            self.parse_type()
        scope = Scope(self, None)
        for p, _ in params:
            scope.declare(p)
        body = self.parse_block(scope, 1)
        args = ", ".join(scope.local(p) for p, _ in params)
        self.py.append(f"def F_{name}(G{', ' if args else ''}{args}):")
        self.py.append("    if False: yield")
        self.py.extend(body if body else ["    pass"])
        self.py.append("")
        self.functions[name] = [p for p, _ in params]
        if any(a[0] == "compute" for a in attrs):
            wg = next(a[1] for a in attrs if a[0] == "workgroup_size")
            self.entries[name] = {"workgroup_size": wg, "builtins": [(p, pa[0][1][0]) for p, pa in params]}

    # -- statements (emit Python lines at indentation `ind`)
    def parse_block(self, scope, ind):
        self.expect("{")
        inner = Scope(self, scope)
        out = []
        while not self.accept("}"):
            out.extend(self.parse_stmt(inner, ind))
        return out

    def parse_simple(self, scope):
        """let / var / assignment / call / ++ without the trailing ';' -> list of python statements (no indentation)."""
        t = self.peek()
        if t in ("let", "var", "const"):
            self.next()
            name = self.next()
            ty = self.parse_type() if self.accept(":") else None
            if self.peek() != "=":  # `var adj: mat2x2<f32>;` -- zero-initialised
                scope.declare(name)
                return [f"{scope.local(name)} = zero_of({ty!r})"]
            self.expect("=")
            rhs = self.parse_expr(scope)
            scope.declare(name)  # after the right-hand side: `let shape = f(shape)` reads the outer `shape`
            return [f"{scope.local(name)} = copyval({rhs})"]
        through_ptr = self.accept("*")  # `*x = ..`: a store through a ptr<function, T>
        lhs = LV(self.parse_postfix(scope).read() + ".v") if through_ptr else self.parse_postfix(scope, lvalue=True)
        t = self.peek()
        if t in ("=", "+=", "-=", "*=", "/="):
            self.next()
            rhs = self.parse_expr(scope)
            if t != "=":
                cur = lhs.read()
                rhs = {"+=": "op_add", "-=": "op_sub", "*=": "op_mul", "/=": "op_div"}[t] + f"({cur}, {rhs})"
            return [lhs.write(f"copyval({rhs})" if through_ptr else rhs)]
        if t in ("++", "--"):
            self.next()
            return [lhs.write(("op_add" if t == "++" else "op_sub") + f"({lhs.read()}, 1)")]
        return [lhs.read()]  # expression statement (a call)

    def parse_stmt(self, scope, ind):
        pad = "    " * ind
        t = self.peek()
        if t == "{":
            return self.parse_block(scope, ind)
        self.pre, self.post = [], []   # `&x` in the statement: cells made before it, written back to the locals after it
        if t == "while":
            self.next()
            cond = self.parse_expr(scope)
            assert not self.pre, "`&x` in a loop condition is not supported"
            self.loops.append(None)
            body = self.parse_block(scope, ind + 1)
            self.loops.pop()
            return [f"{pad}while {cond}:"] + (body or [f"{pad}    pass"])
        if t == "break":
            self.next(); self.expect(";")
            flag = self.loops[-1]
            return [f"{pad}break"] if flag is None else [f"{pad}{flag} = True", f"{pad}break"]
        if t == "if":
            self.next()
            cond = self.parse_expr(scope)
            assert not self.pre, "`&x` in an if condition is not supported"
            out = [f"{pad}if {cond}:"] + (self.parse_block(scope, ind + 1) or [f"{pad}    pass"])
            if self.accept("else"):
                if self.peek() == "if":
                    out += [f"{pad}else:"] + self.parse_stmt(scope, ind + 1)
                else:
                    out += [f"{pad}else:"] + (self.parse_block(scope, ind + 1) or [f"{pad}    pass"])
            return out
        if t == "for":
            self.next()
            self.expect("(")
            inner = Scope(self, scope)
            init = self.parse_simple(inner) if self.peek() != ";" else []
            self.expect(";")
            cond = self.parse_expr(inner) if self.peek() != ";" else "True"
            self.expect(";")
            upd = self.parse_simple(inner) if self.peek() != ")" else []
            self.expect(")")
            assert not self.pre, "`&x` in a for header is not supported"
            self.nflag += 1
            flag = f"_brk{self.nflag}"
            self.loops.append(flag)
            body = self.parse_block(inner, ind + 2)
            self.loops.pop()
            # `continue` must still run the update: the body sits in a one-trip loop and `continue` leaves that one; `break` leaves it with the
            # loop's flag set, and the flag leaves the `while`
            return ([pad + s for s in init] + [f"{pad}{flag} = False", f"{pad}while {cond}:", f"{pad}    for _once in (0,):"] +
                    (body or [f"{pad}        pass"]) + [f"{pad}    if {flag}: break"] + [f"{pad}    {s}" for s in upd])
        if t == "return":
            self.next()
            if self.accept(";"):
                return [f"{pad}return"]
            e = self.parse_expr(scope)
            self.expect(";")
            return [pad + x for x in self.pre] + [f"{pad}return {e}"]
        if t == "continue":
            self.next(); self.expect(";")
            return [f"{pad}break"] if self.loops[-1] is not None else [f"{pad}continue"]
        if t == "workgroupBarrier":
            self.next(); self.expect("("); self.expect(")"); self.expect(";")
            return [f"{pad}yield"]
        body = self.parse_simple(scope)
        out = [pad + s for s in self.pre + body + self.post]
        self.pre, self.post = [], []
        self.expect(";")
        return out

    # -- expressions -> python expression strings
    def parse_expr(self, scope):
        return self.parse_bin(scope, 0)

    LEVELS = [["||"], ["&&"], ["==", "!=", "<", ">", "<=", ">="], [">>"], ["+", "-"], ["*", "/", "%"]]

    def parse_bin(self, scope, lvl):
        if lvl == len(self.LEVELS):
            return self.parse_unary(scope)
        lhs = self.parse_bin(scope, lvl + 1)
        while self.peek() in self.LEVELS[lvl]:
            op = self.next()
            rhs = self.parse_bin(scope, lvl + 1)
            if op in ("+", "-", "*", "/"):
                lhs = {"+": "op_add", "-": "op_sub", "*": "op_mul", "/": "op_div"}[op] + f"({lhs}, {rhs})"
            elif op == "||":
                lhs = f"({lhs} or {rhs})"
            elif op == "&&":
                lhs = f"({lhs} and {rhs})"
            elif op == "%":
                lhs = f"({lhs} % {rhs})"
            elif op == ">>":
                lhs = f"op_shr({lhs}, {rhs})"
            else:
                lhs = f"({lhs} {op} {rhs})"
        return lhs

    def parse_unary(self, scope):
        if self.accept("-"):
            return f"op_neg({self.parse_unary(scope)})"
        if self.accept("&"):  # `&x`: a cell around the local's value, copied back into the local after the statement
            node = self.parse_postfix(scope)
            if node.base is None and node.obj is None and node.expr.startswith("L_"):
                self.nref += 1
                cell = f"_ref{self.nref}"
                self.pre.append(f"{cell} = Ref({node.expr})")
                self.post.append(f"{node.expr} = {cell}.v")
                return cell
            return f"Ref({node.ref()})"  # a component of an array / struct: the object itself is the reference
        if self.accept("*"):
            return self.parse_postfix(scope).read() + ".v"
        if self.accept("!"):
            return f"(not {self.parse_unary(scope)})"
        return self.parse_postfix(scope).read()

    def parse_postfix(self, scope, lvalue=False):
        t = self.next()
        if t == "(":
            e = self.parse_expr(scope)
            self.expect(")")
            node = LV(f"({e})")
        elif re.fullmatch(r"0x[0-9a-fA-F]+u?", t):
            node = LV(str(int(t.rstrip("u"), 16)))
        elif re.fullmatch(r"\d+[ui]", t):
            node = LV(t[:-1])
        elif re.fullmatch(r"\d+", t):
            node = LV(t)
        elif re.fullmatch(r"[\d.]+(?:[eE][+-]?\d+)?f?", t):
            node = LV(f"f32({t.rstrip('f')})")
        else:
            # identifier, possibly namespaced (Alias::name)
            alias = None
            if self.peek() == "::":
                self.next()
                alias, t = t, self.next()
            if self.peek() == "<" and re.fullmatch(r"vec\d|mat\dx\d|array|bitcast", t):  # explicit template arguments: vec4<f32>(...)
                targs = self.parse_type_args()
                if t == "bitcast":
                    t = "bitcast_" + targs[0]
            if self.peek() == "(":
                self.next()
                args = []
                while not self.accept(")"):
                    args.append(self.parse_expr(scope))
                    self.accept(",")
                node = LV(self.call(scope, alias, t, args))
            else:
                node = scope.resolve(alias, t)
        while True:
            if self.accept("."):
                node = LV(None, obj=node, attr=self.next())
            elif self.accept("["):
                idx = self.parse_expr(scope)
                self.expect("]")
                node = LV(None, base=node, index=idx)
            else:
                return node

    def parse_type_args(self):
        self.expect("<")
        depth, toks = 1, []
        while depth:
            x = self.next()
            depth += (x == "<") - (x == ">") - 2 * (x == ">>")
            toks.append(x)
        return toks[:-1]

    def call(self, scope, alias, name, args):
        a = ", ".join(args)
        if alias is None:
            if name in ("vec4", "vec4f") and len(args) in (1, 4):
                return f"fn_vec4({a})"
            if name in ("mat4x4", "mat4x4f") and len(args) in (0, 4):
                return f"fn_mat4x4({a})"
            m = re.fullmatch(r"vec(\d)f?", name)
            if m:
                return f"fn_vec({m.group(1)}{', ' if a else ''}{a})"
            m = re.fullmatch(r"mat(\d)x\1f?", name)
            if m:
                return f"fn_mat({m.group(1)}{', ' if a else ''}{a})"
            if name in ("transpose", "min", "max", "dot", "cross", "sqrt", "length", "sin", "cos", "abs", "sign", "f32", "u32", "atan", "exp", "select",
                        "fma", "bitcast_i32", "bitcast_u32", "bitcast_f32", "array"):
                return f"fn_{name}({a})"
            if name in self.structs:
                if not args:
                    return f"zero_of({name!r})"
                return f"Struct({self.structs[name]!r}, [{a}])"
            name = self.redirect.get(name, name)
            return f"(yield from F_{name}(G{', ' if a else ''}{a}))"
        mod = self.imports[alias]
        if name in mod.structs:
            if not args:
                return f"IMP_{alias}.zero_of({name!r})"
            return f"Struct({mod.structs[name]!r}, [{a}])"
        return f"(yield from IMP_{alias}.ns['F_{name}'](IMP_{alias}.G{', ' if a else ''}{a}))"

    # -- execution
    def run(self, entry: str, grid, bindings: dict):
        """Executes entry point `entry` over grid = (gx, gy, gz) workgroups. `bindings`: uniform structs (dicts or Struct) and
        storage arrays (NumPy float32; arrays of vec4 are passed flat and viewed as (n, 4))."""
        info = self.entries[entry]
        wgs = [self._const(x) for x in info["workgroup_size"]] + [1, 1]
        wx, wy, wz = wgs[0], wgs[1], wgs[2]

        class G:
            pass
        g = G()
        for k, v in self.consts.items():
            setattr(g, k, eval(v, self.ns, {"G": g}))
        storage_views = {}
        for name, (space, ty) in self.globals_.items():
            if "workgroup" in space:
                continue
            v = bindings[name]
            if "uniform" in space:
                sname = ty.split("::")[-1]
                fields = (self.structs.get(sname) or next(m.structs[sname] for m in self.imports.values() if sname in m.structs))
                v = v if isinstance(v, Struct) else Struct(fields, [int(v[f]) for f in fields])
            elif ty.startswith("array<vec4"):
                assert v.dtype == np.float32 and v.size % 4 == 0
                v = v.reshape(-1, 4)
            elif ty.startswith("array<"):
                assert v.dtype == np.float32
            else:  # a single storage scalar: kept as a 1-element array, see LV.write
                assert v.dtype == np.float32 and v.size >= 1
            setattr(g, name, v)
            storage_views[name] = v
        fn = self.ns["F_" + entry]
        for gz in range(grid[2]):
            for gy in range(grid[1]):
                for gx in range(grid[0]):
                    for name, (space, ty) in self.globals_.items():
                        if "workgroup" in space:
                            setattr(g, name, self._alloc_workgroup(ty, g))
                    gens = []
                    for lz in range(wz):
                        for ly in range(wy):
                            for lx in range(wx):
                                vals = {"global_invocation_id": Vec3(gx * wx + lx, gy * wy + ly, gz * wz + lz),
                                        "local_invocation_id": Vec3(lx, ly, lz), "workgroup_id": Vec3(gx, gy, gz)}
                                gens.append(fn(g, *[vals[b] for _, b in info["builtins"]]))
                    live = gens
                    while live:  # advance every invocation to its next barrier (or to its end)
                        nxt = []
                        for it in live:
                            try:
                                next(it)
                                nxt.append(it)
                            except StopIteration:
                                pass
                        live = nxt
        return storage_views

    def _const(self, tok):
        return int(tok) if tok.isdigit() else int(eval(self.consts[tok], self.ns, {}))

    def _alloc_workgroup(self, ty, g):
        m = re.fullmatch(r"array<(.*),(\w+)>", ty)
        elem, n = m.group(1), m.group(2)
        n = int(n) if n.isdigit() else int(getattr(g, n))
        shape = {"f32": (n,), "vec4<f32>": (n, 4), "mat4x4<f32>": (n, 4, 4)}[elem]
        return np.zeros(shape, np.float32)


class LV:
    """An expression that may also be assigned to: a name, `base[index]`, or `obj.member` (swizzle reads on vectors, fields on structs)."""

    def __init__(self, expr, base=None, index=None, obj=None, attr=None, scalar_global=False):
        self.expr, self.base, self.index, self.obj, self.attr, self.scalar_global = expr, base, index, obj, attr, scalar_global

    def ref(self):
        """The object itself (no copy): what an element write or a `&x` argument needs."""
        if self.base is not None:
            return f"{self.base.ref()}[{self.index}]"
        if self.obj is not None:
            return f"member({self.obj.ref()}, {self.attr!r})"
        return self.expr

    def read(self):
        if self.base is not None:
            return f"load({self.base.ref()}, {self.index})"
        if self.obj is not None:
            return f"member({self.obj.ref()}, {self.attr!r})"
        return self.expr + ("[0]" if self.scalar_global else "")

    def write(self, rhs):
        if self.base is not None:
            return f"{self.base.ref()}[{self.index}] = {rhs}"
        if self.obj is not None:
            return f"setmember({self.obj.ref()}, {self.attr!r}, {rhs})"
        if self.scalar_global:
            return f"{self.expr}[0] = {rhs}"
        return f"{self.expr} = {rhs}"


class Scope:
    """Block scope. A name declared again in an inner block (or again in the same one) gets a Python local of its own, so that leaving the block
    gives the outer variable back: `let m = end - 1u;` inside a loop of a function whose matrix is called `m` (eig3.wgsl:82)."""

    def __init__(self, mod, parent):
        self.mod, self.parent, self.names = mod, parent, {}
        self.used = parent.used if parent is not None else set()   # python names taken in this function

    def declare(self, name):
        py, k = "L_" + name, 0
        while py in self.used:
            k += 1
            py = f"L_{name}__{k}"
        self.used.add(py)
        self.names[name] = py

    def local(self, name):
        sc = self
        while sc is not None:
            if name in sc.names:
                return sc.names[name]
            sc = sc.parent
        raise NameError(name)

    def has(self, name):
        return name in self.names or (self.parent is not None and self.parent.has(name))

    def resolve(self, alias, name):
        if alias is not None:
            raise SyntaxError(f"namespaced value {alias}::{name} is not supported")
        if self.has(name):
            return LV(self.local(name))
        if name in self.mod.consts or name in self.mod.globals_:
            space, ty = self.mod.globals_.get(name, ("", ""))
            scalar = name in self.mod.globals_ and "storage" in space and not ty.startswith("array")
            return LV(f"G.{name}", scalar_global=scalar)
        if name in ("true", "false"):
            return LV(name.capitalize())
        raise NameError(f"unknown identifier {name!r}")


def load_linalg(ref_dir: str, file: str, defs: set | None = None, redirect: dict | None = None) -> Module:
    """`ref_dir` = .../crates/wgebra/src/linalg of the reference checkout. Every kernel file imports shape.wgsl as `Shape`."""
    import os
    shape = Module(open(os.path.join(ref_dir, "shape.wgsl")).read(), defs=defs)
    return Module(open(os.path.join(ref_dir, file)).read(), imports={"Shape": shape}, defs=defs, redirect=redirect)
