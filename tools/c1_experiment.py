import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
import wgmath_amd as wg
which = sys.argv[1]
gpu = wg.GpuInstance.new(0)
def graph_us():
    w = bench.WORKLOADS["gemv_f32_1024_graph"]()
    w.setup(wg, gpu, 0, 1)
    for _ in range(20): w.step()
    gpu.sync()
    t0 = time.perf_counter(); n = 300
    for _ in range(n): w.step()
    gpu.sync()
    return (time.perf_counter() - t0) / n / 64 * 1e6
if which == "reserve":
    gpu.device().reserve_workspace(64 << 20) if hasattr(gpu.device(), "reserve_workspace") else gpu.reserve_workspace(64 << 20)
elif which == "biggemv_keep":
    X = bench.WORKLOADS["gemv_f32_4096x65536"](); X.setup(wg, gpu, 0, 1); X.step(); gpu.sync()
elif which == "biggemv_free":
    X = bench.WORKLOADS["gemv_f32_4096x65536"](); X.setup(wg, gpu, 0, 1); X.step(); gpu.sync(); X = None
    import gc; gc.collect()
elif which == "biggemv_nostep":
    X = bench.WORKLOADS["gemv_f32_4096x65536"](); X.setup(wg, gpu, 0, 1); gpu.sync(); X = None
elif which == "midgemv":
    X = bench.GemvWorkload("g", 4096, 8192, False); X.setup(wg, gpu, 0, 1); X.step(); gpu.sync(); X = None
elif which == "gemvtr":
    X = bench.WORKLOADS["gemvtr_f32_65536x4096"](); X.setup(wg, gpu, 0, 1); X.step(); gpu.sync(); X = None
print(f"{which:16s} replay {graph_us():.2f} us; second instance {graph_us():.2f} us", flush=True)
