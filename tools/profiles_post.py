#!/usr/bin/env python3
"""Post-processes the rocprofv3 output of tools/make_profiles.sh into the small CSV/JSON summaries kept under profiles/."""
import csv, glob, json, os, sys, collections, shutil

out = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"

def find(pattern):
    fs = sorted(glob.glob(os.path.join(out, pattern), recursive=True))
    return fs

# 1. kernel stats + durations by (kernel, grid)
stats = find("kt/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats[0], os.path.join(out, f"{tag}_bench_kernel_stats.csv"))
rows = collections.defaultdict(list)
for f in find("kt/**/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        rows[(r["Kernel_Name"], grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(os.path.join(out, f"{tag}_kernel_durations_by_grid.csv"), "w") as fo:
    fo.write("# derived from rocprofv3 --kernel-trace (same run as the round's _bench_kernel_stats.csv: tools/make_profiles.sh step 1).\n")
    fo.write("# mean duration per kernel AND grid size in threads (one kernel name can serve several workloads). Microseconds.\n")
    fo.write("kernel,grid_size,calls,mean_us,min_us,max_us\n")
    for (k, g), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        fo.write(f"\"{k}\",{g},{len(v)},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f}\n")

# (traffic per workload: tools/pmc.sh / tools/pmc_post.py, per-workload counter passes)
