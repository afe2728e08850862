#!/usr/bin/env python3
"""Filler instructions per MFMA-to-MFMA gap in the main loop of an f16 kernel (from `hipcc -S --cuda-device-only` output).
usage: tools/gap_hist.py file.s [ILb0|ILb1]   (NN | TN instance of gemm_f16_m16_kernel)
One wave per SIMD hides ~3 single-issue instructions behind a 16x16x32 MFMA; every further one in the same gap costs ~4.8 cycles
(profiles/r02_evidence.md 3d). tests/test_abi_and_host.py::test_f16_gemm_main_loop_issue_budget pins the shipped kernel to that budget."""
import re
import sys
from collections import Counter


def kernel_text(asm: str, which: str, kernel: str = "gemm_f16_m16_kernel") -> str:
    """The lines of one kernel instance: from its label to .Lfunc_end (a kernel has several s_endpgm)."""
    m = re.search(r"^(_ZN\S*" + kernel + which + r"\S*):.*?^\.Lfunc_end", asm, flags=re.S | re.M)
    if not m:
        raise ValueError(f"{kernel}{which} not found")
    return m.group(0)


def analyse(asm: str, which: str = "ILb0", kernel: str = "gemm_f16_m16_kernel") -> dict:
    K = kernel_text(asm, which, kernel).split("\n")
    headers = [i for i, l in enumerate(K) if "Loop Header" in l]
    out = {"loops": len(headers), "mfma_total": sum("v_mfma" in l for l in K), "spills": sum("Folded" in l for l in K),
           "acc_moves_total": sum(("v_accvgpr_mov" in l) or ("v_accvgpr_write" in l and not l.rstrip().endswith(", 0")) for l in K)}
    best = None  # the loop that holds the most MFMAs
    for h in headers:
        hl = h
        while not K[hl].startswith(".LBB"):  # nested loops: the label sits on the first line of a multi-line loop comment
            hl -= 1
        lbl = K[hl].split(":")[0]
        ends = [i for i, l in enumerate(K) if i > h and "s_cbranch" in l and l.split()[-1] == lbl]
        if ends:
            n = sum("v_mfma" in l for l in K[h:ends[-1]])
            if best is None or n > best[2]:
                best = (h, ends[-1], n)
    hs, be = best[0], best[1]
    body = [l.strip() for l in K[hs + 1:be + 1] if l.strip() and not l.strip().startswith(";") and not l.strip().startswith(".")]
    gaps, cur = [], []
    for l in body:
        if l.startswith("v_mfma"):
            gaps.append(cur)
            cur = []
        else:
            cur.append(l.split()[0])
    gaps[0] = cur + gaps[0]  # the gap across the back-edge
    out.update(loop_instructions=len(body), loop_mfma=len(gaps), gaps=gaps,
               loop_acc_moves=sum(x.startswith("v_accvgpr") for g in gaps for x in g), loop_scratch=sum(x.startswith("scratch_") for g in gaps for x in g),
               excess=sum(max(0, len(g) - 3) for g in gaps))
    return out


if __name__ == "__main__":
    r = analyse(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "ILb0")
    print(f"kernel: {r['loops']} loop(s), {r['mfma_total']} MFMAs in all, {r['spills']} spill instructions, {r['acc_moves_total']} accumulator moves (v_accvgpr_mov / non-zero write)")
    print(f"main loop: {r['loop_instructions']} instructions, {r['loop_mfma']} MFMAs, {r['loop_acc_moves']} accumulator moves, {r['loop_scratch']} scratch accesses")
    print("fillers per gap histogram:", sorted(Counter(len(g) for g in r["gaps"]).items()), " excess over 3:", r["excess"])
    for i, g in enumerate(r["gaps"]):
        print(i, len(g), " ".join(g))
