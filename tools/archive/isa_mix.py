#!/usr/bin/env python3
"""Instruction mix of the steady-state loop of the shipped f16 Gemm kernels, from the ISA hipcc emits (no GPU needed):
   python tools/isa_mix.py > profiles/r02_isa_mix.csv
One row per (kernel variant, opcode) + a summary row: instructions per loop iteration (one stage = 64 k = 128 MFMAs per wave) and
non-MFMA instructions per MFMA (the vendor's best kernel for two k-contiguous operands: ~0.8)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "wgmath_amd", "csrc", "gemm_f16.hip")
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on", "-w", "-save-temps", "-c", src,
                    "-o", os.path.join(d, "g.o"), "-I", os.path.dirname(src)], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(d) if f.endswith(".s") and "gfx950" in f][0]
    lines = open(os.path.join(d, asm)).read().split("\n")
print("kernel,opcode,count_per_loop_iteration")
for name, pat in (("gemm_f16_m16_kernel<TN>", "_ZN5wgf1612_GLOBAL__N_119gemm_f16_m16_kernelILb1"), ("gemm_f16_m16_kernel<NN>", "_ZN5wgf1612_GLOBAL__N_119gemm_f16_m16_kernelILb0")):
    i = [k for k, l in enumerate(lines) if l.startswith(pat)][0]
    j = i
    while not lines[j].startswith(".Lfunc_end"):
        j += 1
    body = lines[i:j]
    labels = {l.split(":")[0]: k for k, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for k, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            seg = body[labels[m.group(1)]:k + 1]
            n = sum("v_mfma" in x for x in seg)
            if best is None or n > best[0]:
                best = (n, seg)
    n, seg = best
    cnt = collections.Counter()
    for x in seg:
        x = x.strip()
        if not x or x.startswith(";") or x.startswith(".") or x.endswith(":"):
            continue
        cnt[x.split()[0]] += 1
    tot = sum(cnt.values())
    for op, c in cnt.most_common():
        print(f"{name},{op},{c}")
    print(f"{name},TOTAL,{tot}")
    print(f"{name},NON_MFMA_PER_MFMA,{(tot - n) / n:.3f}")
