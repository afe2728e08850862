#!/usr/bin/env python3
"""What ONE rank of a P-rank M-sharded f16 Gemm 32768^3 does, emulated on one GPU -- the measured inputs of DESIGN.md section 6's
expected multi-GPU speed-up. Rank 0 of P computes its (M/P x N) row block panel by panel on all 256 CUs (no exchange), then the staged
engine's compute side (Gemm into the staging cube + relayout of every panel, copies switched off: what the per-link SDMA copies run beside;
the link itself is the part a single GPU cannot show). (The SDMA rect-copy engine this tool also emulated up to round 3 was removed in ABI 3.)
Reports, per P: ms per step. Usage: python tools/rank_emulation.py [P ...] > gpurun_out/rank_emulation.json"""
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

N = K = M = int(os.environ.get("EMU_SIZE", "32768"))  # (tests run the same sequence of communicators and contexts on a smaller problem)
Ps = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
STEPS = int(os.environ.get("STEPS", "6"))
out = {"problem": f"f16 {M}^3, M-sharded; one rank emulated on one GPU", "steps": STEPS, "ranks": {}}
T0 = time.time()


def note(*a):
    print(f"[{time.time() - T0:7.2f}s]", *a, file=sys.stderr, flush=True)


import faulthandler  # noqa: E402
import signal  # noqa: E402
# where a run that does not return is standing: every thread's Python stack on stderr shortly before the alarm (the C frame under it is the HIP call that never came back)
faulthandler.dump_traceback_later(max(5, int(os.environ.get("EMU_TIMEOUT", "600")) - 15), exit=False)
signal.alarm(int(os.environ.get("EMU_TIMEOUT", "600")))  # a stuck run ends by itself with SIGALRM (round 5: one 3-rank-count run never returned; see r05_evidence.md section 4)


for P in Ps:
    note(f"[rank_emulation] P = {P}")
    gpu = wg.GpuInstance.new(0)
    dev, S = gpu.device(), wg.BufferUsages
    Mg = M // P
    A = bench.device_random(wg, gpu, (Mg, K), np.float16, 0xA000)
    B = bench.device_random(wg, gpu, (K, N), np.float16, 0xB000)
    C = wg.TensorBuilder.matrix(M, N, S.STORAGE | S.COPY_SRC | S.COPY_DST).build(dev, np.float16)
    note("  operands allocated; creating the communicator")
    comm = Comm(gpu, P, 0, None)
    panel = bench.plan_panel_cols(Mg, N, 256)
    res = {"rows_per_rank": Mg, "panel_cols": panel, "panels": -(-N // panel)}
    for mode, name in ((GatherMode.NONE, "no_exchange"),):
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
    # the staged engine's compute side: the rank's Gemm into the staging cube + the relayout of every panel, copies switched off
    # (WG_STAGED_NO_COPY) and the flags pre-set so that the wait kernels pass: what the contiguous per-link copies run BESIDE.
    # Panel by panel (round 2's form) and as ONE launch per step (round 3), on all 256 CUs; and on a 248-CU masked stream = the compute
    # side of the RCCL engine (the same kernel + the same relayouts; RCCL's own copy kernels run on the 8 CUs left over).
    def staged_side(inst, cm, a_t, b_t, c_t, one_launch, panels=None):
        panels = panels or panel
        cm.set_one_launch(one_launch)
        for _ in range(2):
            cm.sharded_gemm(c_t, a_t, b_t, 0, GatherMode.PEER_STAGED, panels)
        inst.sync()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            cm.sharded_gemm(c_t, a_t, b_t, 0, GatherMode.PEER_STAGED, panels)
        inst.sync()
        return (time.perf_counter() - t0) / STEPS

    def prepare_staged(inst, cm):
        from wgmath_amd._lib import check, lib
        st, fl = cm.stage_reserve(2 * M * N * 2)
        ones = np.full(16 * 1024, 0x3fffffff, np.uint32)
        check(lib.wg_buf_write(inst._ctx.handle, fl, 0, ones.ctypes.data, ones.nbytes))
        cm._stage = (st, fl)
        cm.set_local_peer_stages([cm] * P)

    if P > 1:
        os.environ["WG_STAGED_NO_COPY"] = "1"
        prepare_staged(gpu, comm)
        last = Mg * min(panel, N) * 2
        for one, key in ((False, "staged_compute_and_relayout_panel_launches"), (True, "staged_compute_and_relayout")):
            note("  staged side, one launch:", one)
            dt = staged_side(gpu, comm, A, B, C, one)
            res[key] = {"ms_per_step": round(dt * 1e3, 3), "tflops_of_this_rank": round(2.0 * Mg * N * K / dt / 1e12, 1), "last_panel_bytes_per_peer": last,
                        "one_launch_per_step": one}
    out["ranks"][str(P)] = res
    note("  closing the communicator")
    comm.close()
    if P > 1:  # the RCCL engine's compute side: 248 of the 256 CUs
        note("  creating the 248-CU context")
        gpu2 = wg.GpuInstance.new(0, cu_count=248, one_xcd=os.environ.get("WG_BENCH_CU_MASK_SPREAD") != "1")  # as bench.py does for the RCCL engine
        comm2 = Comm(gpu2, P, 0, None)
        prepare_staged(gpu2, comm2)
        note("  248-CU stream, uniform panels")
        dt = staged_side(gpu2, comm2, A, B, C, True)
        res["rccl_compute_side_248_cus"] = {"ms_per_step": round(dt * 1e3, 3), "tflops_of_this_rank": round(2.0 * Mg * N * K / dt / 1e12, 1), "stream_compute_units": 248}
        # round 5: the same with the tapered tail bench.py now gives that engine (wg_gemm_sharded_panels): more, narrower panels at the end of a step
        from wgmath_amd.sharded import tapered_panels
        widths = tapered_panels(N, panel, float(os.environ.get("WG_BENCH_TAPER", "0.72")))
        note("  248-CU stream, tapered panels", widths)
        dt = staged_side(gpu2, comm2, A, B, C, True, widths)
        res["rccl_compute_side_248_cus_tapered"] = {"ms_per_step": round(dt * 1e3, 3), "tflops_of_this_rank": round(2.0 * Mg * N * K / dt / 1e12, 1),
                                                    "stream_compute_units": 248, "panel_widths": widths}
        note("  closing the 248-CU communicator and context")
        comm2.close()
        gpu2.close()
    del A, B, C
    note("  closing the context")
    gpu.close()
note("[rank_emulation] single-GPU reference run")
# the best single-GPU run: the plain 32768^3 Gemm (one launch, tile scheduler on) -- what every speed-up below is against
gpu = wg.GpuInstance.new(0)
dev, shapes, S = gpu.device(), wg.ViewShapeBuffers(), wg.BufferUsages
A = bench.device_random(wg, gpu, (M, K), np.float16, 0xA000)
B = bench.device_random(wg, gpu, (K, N), np.float16, 0xB000)
C = wg.TensorBuilder.matrix(M, N, S.STORAGE).build(dev, np.float16)
gemm = wg.Gemm.from_device(dev)
enc = dev.create_command_encoder(); ps = enc.compute_pass("t1", None)
for _ in range(2):
    gemm.dispatch(dev, shapes, ps, C, A, B)
gpu.sync()
t0 = time.perf_counter()
for _ in range(STEPS):
    gemm.dispatch(dev, shapes, ps, C, A, B)
gpu.sync()
t1 = (time.perf_counter() - t0) / STEPS * 1e3
out["single_gpu_best"] = {"ms_per_step": round(t1, 3), "tflops": round(2.0 * M * N * K / t1 / 1e9, 1), "what": "wg_gemm 32768^3, one launch, tile scheduler"}
del A, B, C
gpu.close()
SLOW = {2: 1.006, 4: 1.023, 8: 1.048}
ENGINE_GBS = 60.7


def exposed_tail_ms(widths, step_ms, rows, slow):
    """Timeline of one step: panel p's Gemm takes its share of the step's compute time, its exchange (one link at the engine rate) starts when the Gemm
    and the previous panel's exchange are done; what is left of the exchanges when the last Gemm ends is exposed."""
    t_gemm = t_exch = 0.0
    for w in widths:
        t_gemm += step_ms * slow * w / N
        t_exch = max(t_exch, t_gemm) + rows * w * 2 / (ENGINE_GBS * 1e6)
    return t_exch - t_gemm


# measured beside busy SDMA engines (tools/cpp/sdma_probe2.cpp): Gemm slowdown with 1 / 3 / 7 engines at 60.7 GB/s each
if t1:
    for P, r in out["ranks"].items():
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
        if "rccl_compute_side_248_cus" in r:
            rc = r["rccl_compute_side_248_cus"]
            steady = r["rows_per_rank"] * N * 2 / (ENGINE_GBS * 1e6)
            tail = r["staged_compute_and_relayout"]["last_panel_bytes_per_peer"] / (ENGINE_GBS * 1e6)
            t = max(rc["ms_per_step"] * SLOW.get(int(P), 1.05), steady) + tail
            tp = max(rc["ms_per_step"] * SLOW.get(int(P), 1.05), steady)
            r["expected_rccl_248_pipelined"] = {"ms_per_step": round(tp, 3), "speedup_vs_1_gpu": round(t1 / tp, 3),
                                                "note": "wg_comm_set_pipelined: the last panel's gather + relayout run behind the next step's kernel"}
            r["expected_rccl_248"] = {"ms_per_step": round(t, 3), "speedup_vs_1_gpu": round(t1 / t, 3),
                                      "assumes": "RCCL's gather of a panel keeps up with the per-link rate measured for the copy engines (60.7 GB/s) on its 8 CUs; the last panel's gather is exposed"}
            if "rccl_compute_side_248_cus_tapered" in r:
                rt = r["rccl_compute_side_248_cus_tapered"]
                sl = SLOW.get(int(P), 1.05)
                uniform = [r["panel_cols"]] * (N // r["panel_cols"])
                tail_u = exposed_tail_ms(uniform, rc["ms_per_step"], r["rows_per_rank"], sl)
                tail_t = exposed_tail_ms(rt["panel_widths"], rt["ms_per_step"], r["rows_per_rank"], sl)
                tt = max(rt["ms_per_step"] * sl, steady) + tail_t
                r["expected_rccl_248_tapered"] = {"ms_per_step": round(tt, 3), "speedup_vs_1_gpu": round(t1 / tt, 3), "exposed_tail_ms": round(tail_t, 3),
                                                  "exposed_tail_ms_uniform_panels_same_timeline": round(tail_u, 3),
                                                  "assumes": "as expected_rccl_248 (plain, not pipelined); the tail from a per-panel timeline: panel p's gather starts when its Gemm and panel p-1's gather are done"}
print(json.dumps(bench.finite(out), indent=1, allow_nan=False))
