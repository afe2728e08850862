#!/usr/bin/env python3
"""What ONE rank of a P-rank M-sharded f16 Gemm 32768^3 does, emulated on one GPU -- the measured inputs of DESIGN.md section 6's
expected multi-GPU speed-up. Rank 0 of P computes its (M/P x N) row block panel by panel on all 256 CUs; with the peer-copy engine its
SDMA engines push every panel's block into P-1 peer buffers -- here P-1 other buffers of the SAME device, so the local HBM sees the reads
of the outgoing copies plus writes standing in for the incoming ones (the link is the part a single GPU cannot show: a same-device SDMA
rect copy runs at ~60 GB/s, about the xGMI per-direction rate). Reports, per P: ms per step with no exchange, with the exchange, the
exposed tail after the last Gemm, and the copy rate. Usage: python tools/rank_emulation.py [P ...] > gpurun_out/rank_emulation.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np

import bench
import wgmath_amd as wg
from wgmath_amd.sharded import Comm, GatherMode

N = K = M = 32768
Ps = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
STEPS = int(os.environ.get("STEPS", "6"))
out = {"problem": "f16 32768^3, M-sharded; one rank emulated on one GPU", "steps": STEPS, "ranks": {}}
for P in Ps:
    gpu = wg.GpuInstance.new(0)
    dev, S = gpu.device(), wg.BufferUsages
    Mg = M // P
    A = bench.device_random(wg, gpu, (Mg, K), np.float16, 0xA000)
    B = bench.device_random(wg, gpu, (K, N), np.float16, 0xB000)
    C = wg.TensorBuilder.matrix(M, N, S.STORAGE | S.COPY_SRC | S.COPY_DST).build(dev, np.float16)
    npeers = min(P - 1, 3)  # 3 stand-in buffers are enough to keep 3 engines busy; more peers reuse them round-robin
    peers = [wg.TensorBuilder.matrix(M, N, S.STORAGE | S.COPY_DST).build(dev, np.float16) for _ in range(npeers)]
    comm = Comm(gpu, P, 0, None)
    if P > 1:
        comm.register_local_peers(C, [C] + [peers[i % npeers] for i in range(P - 1)])
    panel = bench.plan_panel_cols(Mg, N, 256)
    res = {"rows_per_rank": Mg, "panel_cols": panel, "panels": -(-N // panel), "copy_engine": comm.copy_engine}
    for mode, name in ((GatherMode.NONE, "no_exchange"), (GatherMode.PEER_COPY, "peer_copy")):
        if P == 1 and mode == GatherMode.PEER_COPY:
            continue
        for _ in range(2):
            comm.sharded_gemm(C, A, B, 0, mode, panel)
            gpu.sync(); comm.flush()
        t0 = time.perf_counter()
        tails = []
        for _ in range(STEPS):
            comm.sharded_gemm(C, A, B, 0, mode, panel)
            gpu.sync()
            t1 = time.perf_counter()
            comm.flush()  # the copies still in flight after the last Gemm: the exposed part of the exchange
            tails.append(time.perf_counter() - t1)
        dt = (time.perf_counter() - t0) / STEPS
        res[name] = {"ms_per_step": round(dt * 1e3, 3), "exposed_tail_ms": round(float(np.mean(tails)) * 1e3, 3),
                     "tflops_of_this_rank": round(2.0 * Mg * N * K / dt / 1e12, 1)}
        if mode == GatherMode.PEER_COPY:
            sent = (P - 1) * Mg * N * 2
            res[name]["bytes_pushed_per_step"] = sent
            res[name]["aggregate_push_gbs_if_fully_overlapped"] = round(sent / dt / 1e9, 1)
    # the staged engine's compute side: panel Gemms into the staging cube + the relayout of every panel, copies switched off
    # (WG_STAGED_NO_COPY) and the flags pre-set so that the wait kernels pass: what the contiguous per-link copies run BESIDE
    if P > 1:
        os.environ["WG_STAGED_NO_COPY"] = "1"
        from wgmath_amd._lib import check, lib
        st, fl = comm.stage_reserve(2 * M * N * 2)
        ones = np.full(16 * 1024, 0x3fffffff, np.uint32)
        check(lib.wg_buf_write(gpu._ctx.handle, fl, 0, ones.ctypes.data, ones.nbytes))
        comm._stage = (st, fl)
        comm.set_local_peer_stages([comm] * P)
        for _ in range(2):
            comm.sharded_gemm(C, A, B, 0, GatherMode.PEER_STAGED, panel)
        gpu.sync()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            comm.sharded_gemm(C, A, B, 0, GatherMode.PEER_STAGED, panel)
        gpu.sync()
        dt = (time.perf_counter() - t0) / STEPS
        last = Mg * min(panel, N) * 2
        res["staged_compute_and_relayout"] = {"ms_per_step": round(dt * 1e3, 3), "tflops_of_this_rank": round(2.0 * Mg * N * K / dt / 1e12, 1),
                                              "last_panel_bytes_per_peer": last}
    out["ranks"][str(P)] = res
    comm.close()
    del A, B, C, peers
    gpu.close()
t1 = out["ranks"].get("1", {}).get("no_exchange", {}).get("ms_per_step")
# measured beside busy SDMA engines (tools/cpp/sdma_probe2.cpp): Gemm slowdown with 1 / 3 / 7 engines at 60.7 GB/s each
SLOW = {2: 1.006, 4: 1.023, 8: 1.048}
ENGINE_GBS = 60.7
if t1:
    for P, r in out["ranks"].items():
        if "peer_copy" in r:
            r["expected_speedup_rect_engine"] = round(t1 / r["peer_copy"]["ms_per_step"], 3)
        if "staged_compute_and_relayout" in r:
            sc = r["staged_compute_and_relayout"]
            tail = sc["last_panel_bytes_per_peer"] / (ENGINE_GBS * 1e6)  # ms: the last panel's slot on one link, nothing left to hide it under
            steady = (int(P) - 1) and (r["rows_per_rank"] * N * 2 / (ENGINE_GBS * 1e6))  # ms per step one link needs at the engine rate
            t = max(sc["ms_per_step"] * SLOW.get(int(P), 1.05), steady) + tail
            tp = max(sc["ms_per_step"] * SLOW.get(int(P), 1.05), steady)  # pipelined steps: the tail runs under the next step's first Gemm
            r["expected_staged_pipelined"] = {"ms_per_step": round(tp, 3), "speedup_vs_1_gpu": round(t1 / tp, 3),
                                              "note": "wg_comm_set_pipelined: only the last step of a run exposes its tail"}
            r["expected_staged"] = {"ms_per_step": round(t, 3), "link_ms_per_step_at_engine_rate": round(steady, 3), "exposed_tail_ms": round(tail, 3),
                                    "gemm_slowdown_beside_engines": SLOW.get(int(P), 1.05), "speedup_vs_1_gpu": round(t1 / t, 3)}
print(json.dumps(out, indent=1))
