cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -k "mid or dropped" -m gpu -q 2>&1 | tail -4
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c1prof -o c1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload gemv_f32_1024 --steps 3000 --warmup 200 --no-secondary --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/c1_eager.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c1profg -o c1g -- python3 $GRAFT_REPO_ROOT/bench.py --workload gemv_f32_1024_graph --steps 100 --warmup 10 --no-secondary --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/c1_graph.json 2> /dev/null
cd $GRAFT_REPO_ROOT
find gpurun_out/c1prof gpurun_out/c1profg -name "*kernel_stats.csv" | xargs -I{} sh -c 'echo {}; head -5 {}'
python3 - <<'P'
import csv,glob,statistics
for d in ("c1prof","c1profg"):
    f=glob.glob(f"gpurun_out/{d}/**/*kernel_trace.csv",recursive=True)[0]
    rows=[r for r in csv.DictReader(open(f)) if "gemv_n_small" in r["Kernel_Name"]]
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    dur=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
    gaps=[(int(b["Start_Timestamp"])-int(a["End_Timestamp"]))/1e3 for a,b in zip(rows,rows[1:])]
    print(d, len(rows), "kernel us median", statistics.median(dur), "gap us median", statistics.median(gaps), "grid", rows[0]["Grid_Size_X"] if "Grid_Size_X" in rows[0] else rows[0].get("Grid_Size"), rows[0].get("Workgroup_Size_X"))
P
