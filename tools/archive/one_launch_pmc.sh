#!/bin/bash
# L2 hit rate / fabric traffic of the rank's Gemm as 16 panel launches vs ONE launch (P = 4 compute side): does the long launch lose L2 sharing?
mkdir -p gpurun_out/olpmc; cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/olpmc
cat > /tmp/olpmc.txt <<'PMC'
pmc: TCC_HIT_sum TCC_MISS_sum
pmc: FETCH_SIZE
pmc: GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES
PMC
rocprofv3 -i /tmp/olpmc.txt --kernel-trace --output-format csv -d $O/run -o p -- python3 $GRAFT_REPO_ROOT/tools/one_launch_ab.py 4 > $O/ab.txt 2> $O/ab.log
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/olpmc/run/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        if "gemm_f16_m16" not in r["Kernel_Name"]: continue
        g = int(r["Grid_Size"])
        agg[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
        k = (r["Dispatch_Id"], f)
        if k not in seen:
            seen.add(k); dur[g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for g in sorted(agg):
    m = {c: sum(v) / len(v) for c, v in agg[g].items()}
    hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print(f"grid {g:8d} threads ({g // 256} workgroups): launches {len(dur[g])}, mean {sum(dur[g]) / len(dur[g]):9.1f} us, L2 hit {hit / (hit + miss + 1e-9):.4f}, "
          f"fetch {m.get('FETCH_SIZE', 0) * 2048 / 1e6:9.1f} MB, MFMA busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / (cyc + 1e-9):.3f}, wait_any/wave_cycles {m.get('SQ_WAIT_ANY', 0) / (m.get('SQ_WAVE_CYCLES', 1)):.3f}")
PY
