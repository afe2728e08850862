cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/vk -o vk -- python3 $GRAFT_REPO_ROOT/tools/vendor_kernel_name.py 8192x8192x512 8192x8192x1024 8192x8192x2048 4096x4096x1024 4096x4096x4096 2>&1 | grep vendor
cd $GRAFT_REPO_ROOT
python3 - <<'P'
import csv,glob,collections
f=glob.glob("gpurun_out/vk/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[(r["Kernel_Name"][:150], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X"), r.get("LDS_Block_Size"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items():
    if len(v) >= 20: print(len(v), "median us %.1f" % sorted(v)[len(v)//2], k)
P
