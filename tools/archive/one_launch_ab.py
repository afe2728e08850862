#!/usr/bin/env python3
"""One rank's compute side of the staged engine at P = 4 (f16 32768^3: 8192 x 32768 x 32768 into the staging cube + relayouts, copies off):
panel-by-panel launches vs ONE launch per step. usage: [WGEBRA_HIP_LIB=...] python tools/one_launch_ab.py [P] [cus]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["WG_STAGED_NO_COPY"] = "1"
import numpy as np
import bench
import wgmath_amd as wg
from wgmath_amd.sharded import Comm, GatherMode
from wgmath_amd._lib import check, lib
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cus = int(sys.argv[2]) if len(sys.argv) > 2 else None
N = K = M = 32768
Mg = M // P
gpu = wg.GpuInstance.new(0, cu_count=cus, one_xcd=os.environ.get("WG_BENCH_CU_MASK_SPREAD") != "1") if cus else wg.GpuInstance.new(0)
dev, S = gpu.device(), wg.BufferUsages
A = bench.device_random(wg, gpu, (Mg, K), np.float16, 0xA000); B = bench.device_random(wg, gpu, (K, N), np.float16, 0xB000)
C = wg.TensorBuilder.matrix(M, N, S.STORAGE | S.COPY_SRC | S.COPY_DST).build(dev, np.float16)
comm = Comm(gpu, P, 0, None)
st, fl = comm.stage_reserve(2 * M * N * 2)
ones = np.full(16 * 1024, 0x3fffffff, np.uint32)
check(lib.wg_buf_write(gpu._ctx.handle, fl, 0, ones.ctypes.data, ones.nbytes))
comm._stage = (st, fl); comm.set_local_peer_stages([comm] * P)
panel = bench.plan_panel_cols(Mg, N, cus or 256)
for rep in range(3):
    for one in (False, True):
        comm.set_one_launch(one)
        for _ in range(2): comm.sharded_gemm(C, A, B, 0, GatherMode.PEER_STAGED, panel)
        gpu.sync(); t0 = time.perf_counter()
        for _ in range(10): comm.sharded_gemm(C, A, B, 0, GatherMode.PEER_STAGED, panel)
        gpu.sync(); dt = (time.perf_counter() - t0) / 10
        print(f"P={P} cus={cus or 256} panel={panel} one_launch={one}: {dt*1e3:8.3f} ms  {2.0*Mg*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)
