#!/usr/bin/env python3
"""Config 1 replayed (the 1024 x 1024 Gemv as 64 nodes of a recorded command buffer) comes out at 2.6 us per dispatch in some bench runs and 3.45-3.5 in others -- on the SAME
chip (profiles/r06_evidence.md section 7). This tool finds what in a process's history moves it: a fresh child per candidate workload X runs X for a few steps and then the
replayed Gemv in the same process / context, and prints the replay's us per dispatch.  Usage (GPU box): python tools/c1_population.py [X ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (names only: nothing here touches the GPU)

GRAPH = "gemv_f32_1024_graph"
cands = sys.argv[1:] or ["gemv_f32_1024", "op_assign_f32_256M", "reduce_f32_4096x65536", "gemv_f32_4096x65536", "gemm_f32_4096", "gemm_f32_2048", "gemm_f16_2048",
                         "gemm_f16_8192", "gemmtr_f16_8192", "gemm_f16_32768", "gemmtr_rm_f16_8192", "gemm_f32_fewcols_32000x16x4096", "gemv_f32_4096x65536_rhs8"]
skip = ",".join(n for n in bench.SECONDARY if n != GRAPH)
for x in cands:
    side = f"/tmp/c1pop_{x}.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", x, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--skip", skip, "--detail", side],
                       capture_output=True, text=True, env=dict(os.environ, WG_BENCH_NO_CHECK="1", WG_BENCH_NO_CEILING="1"), timeout=600)
    try:
        d = json.load(open(side))
        g = [o for o in d["others"] if o.get("workload") == GRAPH][0]
        print(f"after {x:34s}: replayed Gemv 1024^2 {g['roofline'].get('dispatch_us')} us per dispatch", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"after {x}: failed ({e}); rc {r.returncode}; {r.stderr[-300:]}", flush=True)
