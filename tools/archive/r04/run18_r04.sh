cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_kw
cd /tmp && export TMPDIR=/tmp
export WG_F32_MID=64064
rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_passes.txt --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_kw/a -o p -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py f32 n 1024x1024x1024 2048x2048x2048 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'P'
import csv,glob,collections
rows=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_kw/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mid_kw" not in r["Kernel_Name"]: continue
        rows[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g,d in rows.items():
    print("grid", g)
    for k,v in sorted(d.items()):
        print("   %-28s mean %.4g (n=%d)" % (k, sum(v)/len(v), len(v)))
P
