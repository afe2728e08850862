#!/usr/bin/env python3
"""Post-processes the rocprofv3 output of tools/make_profiles.sh into the small CSV/JSON summaries kept under profiles/."""
import csv, glob, json, os, sys, collections, shutil

out = sys.argv[1]

def find(pattern):
    fs = sorted(glob.glob(os.path.join(out, pattern), recursive=True))
    return fs

# 1. kernel stats + durations by (kernel, grid)
stats = find("kt/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats[0], os.path.join(out, "r02_bench_kernel_stats.csv"))
rows = collections.defaultdict(list)
for f in find("kt/**/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        rows[(r["Kernel_Name"], grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(os.path.join(out, "r02_kernel_durations_by_grid.csv"), "w") as fo:
    fo.write("# derived from rocprofv3 --kernel-trace (same run as r02_bench_kernel_stats.csv: tools/make_profiles.sh step 1).\n")
    fo.write("# mean duration per kernel AND grid size in threads (one kernel name can serve several workloads). Microseconds.\n")
    fo.write("kernel,grid_size,calls,mean_us,min_us,max_us\n")
    for (k, g), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        fo.write(f"\"{k}\",{g},{len(v)},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f}\n")

# 2. traffic per workload: counters averaged over the launches of the workload's dominant kernel at its largest grid
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in find("pmc/**/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        grid = int(r["Grid_Size"]) if "Grid_Size" in r and r["Grid_Size"] else 0
        agg[(r["Kernel_Name"], grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
def pick(substr, nth_largest_grid=0, exclude=None, grid=None):
    """Counters of the kernel whose name contains `substr`, at an exact grid size in threads (`grid`: one kernel serves several
    workloads of bench.py) or at its n-th largest grid."""
    ks = sorted({k for k in agg if substr in k[0] and not (exclude and exclude in k[0])}, key=lambda k: -k[1])
    grids = sorted({k[1] for k in ks}, reverse=True)
    if grid is not None:
        if grid not in grids: return None
        g = grid
    else:
        if len(grids) <= nth_largest_grid: return None
        g = grids[nth_largest_grid]
    m = collections.defaultdict(list)
    for k in ks:
        if k[1] == g:
            for c, v in agg[k].items(): m[c] += v
    return {c: sum(v) / len(v) for c, v in m.items()}
work = {
    # grid sizes in threads: tiles x 256 (f16: 256 x 256 tiles; f32: 256 x 128 tiles)
    "gemm_f16_32768": pick("gemm_f16_m16_kernel<false>", grid=16384 * 256) or pick("gemm_f16_m16_kernel<false>", 0),
    "gemm_f16_8192": pick("gemm_f16_m16_kernel<false>", grid=1024 * 256) or pick("gemm_f16_m16_kernel<false>", 2),
    "gemm_f16_ts_131072x1024x8192": pick("gemm_f16_m16_kernel<false>", grid=2048 * 256),
    "gemmtr_f16_8192": pick("gemm_f16_m16_kernel<true>", grid=1024 * 256),
    "gemm_f16_2048": pick("gemm_f16_t128_kernel<false>", grid=256 * 256),
    "gemm_f32_4096": pick("gemm_f32_kernel<false>", grid=512 * 256) or pick("gemm_f32", grid=512 * 256),
    "gemm_f32_ts_65536x512x4096": pick("gemm_f32_kernel<false>", grid=1024 * 256) or pick("gemm_f32", grid=1024 * 256),
    "gemv_f32_4096x65536": pick("gemv_n_kernel", 0),
    "gemvtr_f32_65536x4096": pick("gemv_t_kernel", 0),
    "reduce_f32_4096x65536": pick("reduce_rows4", 0),
    "op_assign_f32_256M": pick("op_assign_f32_vec", 0),
}
traffic = {}
for w, m in work.items():
    if not m or "FETCH_SIZE" not in m: continue
    fetch = m["FETCH_SIZE"] * 1024 * 2  # KB -> bytes, x2: gfx950 correction for 16-B/lane streams (MI355X_MICROARCH.md, HBM/rocprofv3)
    write = m.get("WRITE_SIZE", 0.0) * 1024
    hit, miss = m.get("TCC_HIT_sum", 0.0), m.get("TCC_MISS_sum", 0.0)
    traffic[w] = {"hbm_bytes_per_launch": round(fetch + write), "fetch_bytes_corrected": round(fetch), "write_bytes": round(write),
                  "FETCH_SIZE_kb_raw": round(m["FETCH_SIZE"], 1), "WRITE_SIZE_kb_raw": round(m.get("WRITE_SIZE", 0.0), 1),
                  "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None}
traffic["_note"] = ("rocprofv3 --pmc, separate passes (tools/pmc_traffic.txt), mean over the launches of "
                    "`python3 bench.py --steps 4 --warmup 1 --secondary-seconds 0.05` (tools/make_profiles.sh step 2). traffic = FETCH_SIZE*1024*2 + "
                    "WRITE_SIZE*1024: FETCH_SIZE on gfx950 reports half the bytes of a 16-B/lane stream (guide) and counts L2 misses incl. those "
                    "served by the 256 MiB Infinity Cache, so GEMM 'traffic' above the operand size is MALL traffic, not HBM over-fetch.")
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps({k: v.get("hbm_bytes_per_launch") if isinstance(v, dict) else None for k, v in traffic.items()}))
