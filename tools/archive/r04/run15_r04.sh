cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/few -o few -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py f32 t 64x4096x4096 4096x64x4096 64x11008x4096 2>&1 | grep f32
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fewn -o fewn -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py f32 n 64x4096x4096 2>&1 | grep f32
cd $GRAFT_REPO_ROOT
python3 - <<'P'
import csv,glob,collections
for d in ("few","fewn"):
    f=glob.glob(f"gpurun_out/{d}/**/*kernel_trace.csv",recursive=True)[0]
    rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
    agg=collections.defaultdict(list)
    for r in rows:
        agg[(r["Kernel_Name"][:90], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Workgroup_Size_X"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for k,v in agg.items():
        print(d, len(v), "median us %.2f" % sorted(v)[len(v)//2], k)
P
SPLITS="0 4 8 12 16 24 32" timeout 300 python tools/f32_mid_sweep.py 64x4096x4096 4096x64x4096 2>&1 | grep -v "^W2026\|^E2026" | tail -12
