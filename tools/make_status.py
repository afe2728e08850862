#!/usr/bin/env python3
"""Writes STATUS.md -- ONE page of current numbers, nothing historical -- from the bench lines of the current round.

    python tools/make_status.py profiles/r06_bench*_detail.json [--vendor profiles/r06_vendor_vs_ours.txt] > STATUS.md

Every figure comes out of a file named in the page itself: a bench line is the JSON `python bench.py` printed on one box (one file = one box's
run), the vendor column out of tools/vendor_vs_ours.sh's text (lines `<workload> ... ours <x> TF ... vendor <y> TF`). Re-run after every bench run."""
import json
import os
import re
import sys

ROWS = [  # (BASELINE config, workload, label, unit key)
    ("C1", "gemv_f32_1024", "Gemv f32 1024 x 1024, one eager dispatch", "us"),
    ("C1", "gemv_f32_1024_graph", "... replayed from a recorded command buffer (64 per graph)", "us"),
    ("C2", "gemm_f32_4096", "f32 Gemm 4096^3", "TFLOP/s"),
    ("C2", "gemm_f32_4096_u01", "... operands U[0,1) (the reference's `new_random`)", "TFLOP/s"),
    ("C3", "gemm_f16_8192", "f16 Gemm 8192^3", "TFLOP/s"),
    ("C3", "gemmtr_f16_8192", "f16 GemmTr 8192^3", "TFLOP/s"),
    ("C3", "gemm_f16_8192_u01", "f16 Gemm 8192^3, operands U[0,1)", "TFLOP/s"),
    ("C3", "gemmtr_f16_8192_u01", "f16 GemmTr 8192^3, operands U[0,1)", "TFLOP/s"),
    ("C3", "gemmtr_rm_f16_8192", "f16 GemmTr 8192^3 on ROW-major views (`wg_gemm_rm`: both operands contiguous along their output dimension)", "TFLOP/s"),
    ("C4", "gemv_f32_4096x65536", "Gemv f32 4096 x 65536", "GB/s"),
    ("C4", "gemvtr_f32_65536x4096", "GemvTr f32 65536 x 4096", "GB/s"),
    ("C4", "gemv_f32_4096x65536_rhs8", "Gemv f32 4096 x 65536, 8 right-hand sides", "GB/s"),
    ("C4", "reduce_f32_4096x65536", "Reduce (Sum) 4096 vectors of 65536", "GB/s"),
    ("C4", "op_assign_f32_256M", "OpAssign (Add) 2^26 elements", "GB/s"),
    ("C5", "gemm_f16_32768", "f16 Gemm 32768^3 (1 GPU)", "TFLOP/s"),
    ("C5", "gemmtr_f16_32768", "f16 GemmTr 32768^3 (1 GPU)", "TFLOP/s"),
    ("C5", "gemm_f16_32768_u01", "f16 Gemm 32768^3, operands U[0,1)", "TFLOP/s"),
    ("-", "gemm_f16_8192x8192x1024", "f16 Gemm 8192 x 8192 x 1024 (short K: the continuous tile walk)", "TFLOP/s"),
    ("-", "gemmtr_f16_8192x8192x1024", "f16 GemmTr 8192 x 8192 x 1024", "TFLOP/s"),
    ("-", "gemm_f16_2048", "f16 Gemm 2048^3", "TFLOP/s"),
    ("-", "gemm_f32_2048", "f32 Gemm 2048^3", "TFLOP/s"),
    ("-", "gemm_f16_ts_131072x1024x8192", "f16 Gemm 131072 x 1024 x 8192 (tall-skinny)", "TFLOP/s"),
    ("-", "gemm_f32_ts_65536x512x4096", "f32 Gemm 65536 x 512 x 4096 (tall-skinny)", "TFLOP/s"),
    ("-", "gemm_f32_fewcols_32000x16x4096", "f32 Gemm 32000 x 16 x 4096 (few columns, HBM-bound)", "GB/s"),
]


def load_record(path):
    """A bench record: the FULL one -- bench.py's --detail sidecar (or its `[bench detail] {...}` stderr copy) -- carries "others"; a file holding only
    the compact stdout line is completed from the sidecar next to it (`<name>_detail.json`) when there is one."""
    with open(path) as fh:
        txt = [ln for ln in fh.read().strip().splitlines() if ln.strip()]
    last = txt[-1]
    if "[bench detail] {" in last:
        last = last[last.index("{"):]
    rec = json.loads(last)
    if "others" not in rec:
        side = os.path.splitext(path)[0] + "_detail.json"
        if os.path.exists(side):
            rec = json.loads(open(side).read().strip().splitlines()[-1])
    return rec


def by_workload(line):
    out = {line["config"]["workload"]: {"value": line["value"], "roofline": line["roofline"]}}
    for o in line.get("others", []):
        if "roofline" in o:
            out[o["workload"]] = o
    return out


def cell(o, unit):
    if o is None:
        return "-"
    rf = o["roofline"]
    if unit == "us":
        return f"{rf.get('dispatch_us', float('nan')):.2f} us"
    s = f"{o['value']:.0f} ({rf['frac']:.3f}"
    if "frac_of_ceiling" in rf:
        s += f"; {rf['frac_of_ceiling']:.3f} of ceiling"
    s += ")"
    if "clock_ghz_measured" in rf:
        s += f" @ {rf['clock_ghz_measured']:.2f} GHz"
    return s


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    vendor_file = None
    if "--vendor" in sys.argv:
        vendor_file = sys.argv[sys.argv.index("--vendor") + 1]
        args = [a for a in args if a != vendor_file]
    lines = []
    for f in args:
        lines.append((os.path.relpath(f), load_record(f)))
    vendor = {}
    if vendor_file and os.path.exists(vendor_file):
        # tools/vendor_vs_ours.sh: "vendor f16 8192^3 nn: X TFLOP/s" (hipBLASLt through torch.matmul, same box, same moment) and "ours <workload>: Y"
        ven, ours = {}, {}
        for ln in open(vendor_file):
            m = re.match(r"vendor f16 (\d+)\^3 (nn|nt):\s+([\d.]+)", ln)
            if m:
                ven[(int(m.group(1)), m.group(2))] = float(m.group(3))
            m = re.match(r"ours (\S+):\s+([\d.]+)", ln)
            if m:
                ours[m.group(1)] = float(m.group(2))
        for wl, key in (("gemm_f16_8192", (8192, "nn")), ("gemmtr_f16_8192", (8192, "nt")), ("gemm_f16_32768", (16384, "nn"))):
            if wl in ours and key in ven:
                vendor[wl] = (ven[key], ours[wl], key)
    print("# STATUS -- current numbers, one page\n")
    print("Generated by `tools/make_status.py` from the bench lines below (one file = one MI355X box's `python bench.py`); nothing here is historical.")
    print("Cell = value (fraction of the nominal roofline: 2.5 PF f16 / 157.3 TF f32 MFMA, 8 TB/s HBM; for f16 also the fraction of the MFMA-only")
    print("power ceiling measured in the same run) @ the shader clock measured during the timed steps. Operands U[-1,1) unless the row says U[0,1).\n")
    for f, ln in lines:
        c = ln["config"]
        ceil = ln["targets"].get("mfma_only_ceiling_tflops")
        print(f"* `{f}`: {c.get('device', '?')}, {c.get('compute_units', '?')} CUs; MFMA-only ceiling {ceil} TFLOP/s at "
              f"{ln['targets'].get('mfma_only_ceiling_ghz')} GHz; headline `{c['workload']}` {ln['value']:.1f} {ln['unit']}"
              + (f"; box: chip id {c.get('chip_id')}, VBIOS {c.get('vbios')}, firmware MEC {c.get('fw_mec')} / SDMA {c.get('fw_sdma')} / SMC {c.get('fw_smc')}, kernel {c.get('kernel')}"
                 if "chip_id" in c else ""))
    legend = [("before_w16", "before the few-column f32 kernel got its 16-wide MFMA form (few-column row 0.69-0.70 there)"),
              ("before_walk", "before the continuous tile walk (f16 8192^3 and the K = 1024 rows on the per-tile launch there)"),
              ("before_nt", "before the non-temporal hint on the few-column kernel's streamed pieces (few-column row 0.73-0.74 there)")]
    for tag, what in legend:
        if any(tag in f for f, _ in lines):
            print(f"* files named `*_{tag}` were taken earlier in the round, on other boxes, {what}; by code nothing else differs: they are here for the")
            print("  box-to-box spread of the power-capped f16 rows (MFMA-only ceiling 1875-2019 TFLOP/s over the round's boxes).")
    print()
    hdr = "| cfg | workload | " + " | ".join(f"`{os.path.basename(f)}`" for f, _ in lines) + (" | vendor vs ours, same box and moment |" if vendor else " |")
    print(hdr)
    print("|" + "---|" * (hdr.count("|") - 1))
    tables = [by_workload(ln) for _, ln in lines]
    for cfg, wl, label, unit in ROWS:
        cells = [cell(t.get(wl), unit) for t in tables]
        if all(c == "-" for c in cells):
            continue
        row = f"| {cfg} | {label} [{unit}] | " + " | ".join(cells) + " |"
        if vendor:
            if wl in vendor:
                v, o, key = vendor[wl]
                row += f" hipBLASLt {key[0]}^3 {key[1]} {v:.0f} vs {o:.0f} ({o / v:.2f} x) |"
            else:
                row += " |"
        print(row)
    print()
    ck = lines[0][1].get("checks", {})
    if ck:
        print("Parity sampled inside the bench run (|gpu - f64| in ulps; bounds asserted in `tests/test_gpu_ulp.py`): " +
              ", ".join(f"{k.replace('parity_max_ulp_vs_f64_', '')} {v}" for k, v in ck.items() if k.startswith("parity_max")) + ".")
    cpu = lines[0][1].get("cpu_baseline")
    if cpu:
        print(f"\nCPU baseline of the headline (the oracle port on the box's host cores): {cpu['value']:.3g} {cpu['unit']} on {cpu['cores']} threads ({cpu['sample']}).")
    if vendor:
        print(f"\nVendor column: `{vendor_file}` (`tools/vendor_vs_ours.sh`: hipBLASLt through `torch.matmul` on this repo's layouts -- nn = both operands as the Gemm takes them,")
        print("nt = its best layout, what GemmTr computes; the vendor's 32768^3 does not fit its workspace here, 16384^3 stands in). Shape sweeps against the vendor library:")
        print("`profiles/r06_gemm_sweep_full.txt` (116 Gemm / GemmTr cases on the final code: one 10 % behind, f32 GemmTr 64 x 4096 x 4096), `profiles/r06_gemm_sweep_row_major.txt`")
        print("(the row-major surface, 32 cases: f16 GemmTr 0.60-1.07 x the vendor's time, Gemm 0.61-0.94 x; behind: f32 GemmTr 2048^3 1.14 x, on the transposed-copy path below a round of tiles),")
        print("`profiles/r06_misc_sweep.txt` (50 batched / multi-RHS cases: 6 behind by 12-17 %); 48 random shapes x Gemm / GemmTr (`r06_gemm_sweep_random48_*.txt`, `tools/random48_shapes.txt`; f16 / f32 re-run on the final code): f16 16 of 96 behind,")
        print("f32 5 of 96 (16 before the batch-wide tail split), row-major f16 5 of 96 -- mid-size outputs, where the vendor's stream-K kernels fill the chip better.")
    print("\nC1 replayed: the figure depends on what the process allocated before, not on the box -- the same chip gives 2.6 us per dispatch in a fresh process and 3.4-5.7 us behind 25 other")
    print("workloads (`profiles/r06_c1_replay_populations.txt`). The bench therefore runs config 1 right behind the headline since `r06_bench_boxF` (2.65 us); the older files ran it last.")
    print("\nNorth-star targets: f16 Gemm 8192^3 >= 0.80 of MFMA peak -- NOT met (see the C3 rows; the matrix cores alone, on random operands, sustain the")
    print("`MFMA-only ceiling` above at the package power cap); Gemv >= 0.70 of HBM peak -- met (C4); >= 3.5x at 4 GPUs -- not measured on hardware")
    print("(one GPU per box here; `profiles/r06_rank_emulation.json` holds the one-rank emulation the projection in DESIGN.md section 6 rests on).")


if __name__ == "__main__":
    main()
