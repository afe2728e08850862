#!/usr/bin/env python3
"""rocprofv3 counter CSVs of tools/pmc.sh -> <round>_pmc.csv: one row per workload for its dominant kernel (the one with the most GPU
time in that run), counters averaged per launch. Derived columns follow /opt/skills/guides/MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is
summed over the 8 XCDs (cycles per XCD = /8); effective clock = cycles per XCD / kernel duration; MFMA utilisation =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles per XCD); FETCH_SIZE (KiB) is doubled (gfx950 16-B/lane correction) and counts L2
misses including those the Infinity Cache serves."""
import collections
import csv
import glob
import os
import sys

out, tag, wls = sys.argv[1], sys.argv[2], sys.argv[3:]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # algorithmic bytes per launch of every workload (what the measured traffic is compared with)
cols = ["workload", "kernel", "launches", "mean_us", "cycles_per_xcd", "clock_ghz", "mfma_busy_cycles_per_simd", "mfma_util", "sq_busy_cycles", "wave_cycles_x4",
        "wait_inst_any_x4", "active_inst_any_x4", "active_inst_lds_x4", "lds_bank_conflict", "lds_idx_active", "insts_lds", "insts_vmem", "insts_valu",
        "insts_salu", "waves", "tcc_hit", "tcc_miss", "l2_hit_rate", "fetch_bytes_corrected", "write_bytes", "traffic_bytes", "algorithmic_bytes", "traffic_over_algorithmic"]
rows = []
for wl in wls:
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, wl, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            per_kernel[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            did = (r.get("Dispatch_Id"), f)
            if did not in seen:
                seen.add(did)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    cand = {k: sum(v) for k, v in dur.items() if not any(s in k for s in ("Memset", "memset", "fill", "Copy", "copy", "now_kernel", "mfma_only_kernel", "clock_stamp_kernel"))}
    if not cand:
        continue
    k = max(cand, key=cand.get)
    m = {c: sum(v) / len(v) for c, v in per_kernel[k].items()}
    d = sum(dur[k]) / len(dur[k])
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0
    hit, miss = m.get("TCC_HIT_sum", 0.0), m.get("TCC_MISS_sum", 0.0)
    fetch, write = m.get("FETCH_SIZE", 0.0) * 1024 * 2, m.get("WRITE_SIZE", 0.0) * 1024
    g = lambda c: round(m[c], 1) if c in m else ""
    rows.append([wl, k.split("(")[0][:80], len(dur[k]), round(d, 2), round(cyc), round(cyc / d / 1e3, 3) if d else "", round(busy), round(busy / cyc, 4) if cyc else "",
                 g("SQ_BUSY_CYCLES"), g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY"), g("SQ_ACTIVE_INST_ANY"), g("SQ_ACTIVE_INST_LDS"), g("SQ_LDS_BANK_CONFLICT"),
                 g("SQ_LDS_IDX_ACTIVE"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM"), g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_WAVES"), g("TCC_HIT_sum"), g("TCC_MISS_sum"),
                 round(hit / (hit + miss), 4) if hit + miss else "", round(fetch), round(write), round(fetch + write)])
    try:  # a measured byte count below the compulsory traffic cannot be real: refuse to write such a row
        w_ = bench.WORKLOADS[wl]()
        w_.Mg, w_.npanels = getattr(w_, "M", 0), 1
        alg = float(w_.algorithmic_bytes())
        rows[-1] += [round(alg), round((fetch + write) / alg, 3) if alg else ""]
        if alg and fetch + write < 0.98 * alg:
            raise SystemExit(f"{wl}: measured traffic {fetch + write:.4g} B is below 0.98 x the algorithmic {alg:.4g} B -- wrong kernel / grid picked or a counter pass failed")
    except KeyError:
        rows[-1] += ["", ""]
with open(os.path.join(out, f"{tag}_pmc.csv"), "w", newline="") as fo:
    fo.write("# rocprofv3 -i tools/pmc_passes.txt --kernel-trace -- python3 bench.py --steps 6 --warmup 2 --workload W --no-secondary --no-cpu-baseline (tools/pmc.sh)\n")
    fo.write("# per launch of the dominant kernel; profiled passes clock lower than unprofiled runs (guide: never compare the two); *_x4: SQ quad-cycle counters\n")
    w = csv.writer(fo)
    w.writerow(cols)
    w.writerows(rows)
print(open(os.path.join(out, f"{tag}_pmc.csv")).read())
