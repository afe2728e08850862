"""f32 Gemm / GemmTr on mid-size outputs: gemm_f32.hip's plan (knob 0), every tile of the mid family forced, the launcher's own choice (-1) and the
vendor library (torch.matmul on the same memory layout). GPU time per call from back-to-back launches.
Usage (GPU box): python tools/f32_mid_sweep.py [MxNxK[xB] ...]   (torch first: one HIP runtime per process)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import wgmath_amd as wg  # noqa: E402
from bench import device_random  # noqa: E402

gpu = wg.GpuInstance.new(0)
dev, shapes = gpu.device(), wg.ViewShapeBuffers()
S = wg.BufferUsages
gemm = wg.Gemm.from_device(dev)
SHAPES = [(512, 512, 512, 1), (768, 768, 768, 1), (1024, 1024, 1024, 1), (1000, 1000, 1000, 1), (1536, 1536, 1536, 1), (2048, 2048, 2048, 1), (2560, 2560, 2560, 1),
          (3072, 3072, 3072, 1), (4096, 4096, 4096, 1), (1024, 1024, 4096, 1), (4096, 1024, 1024, 1), (1024, 4096, 1024, 1), (2048, 2048, 512, 1),
          (512, 512, 512, 8), (256, 256, 256, 64), (128, 128, 128, 256), (1024, 1024, 1024, 4), (5120, 5120, 2048, 1), (4096, 11008, 4096, 1)]
if sys.argv[1:]:
    SHAPES = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
    SHAPES = [s if len(s) == 4 else s + (1,) for s in SHAPES]
KNOBS = [0, 128128, 128064, 64128, 64064, 64032, 32064, 96096, 96064, -1]
if os.environ.get("SPLITS"):  # few tiles, long K: the 64 x 64 k-split tile with K cut across workgroups -- SPLITS="0 2 4 8" (knob f32_mid_split)
    KNOBS = [0] + [(64064, int(x)) for x in os.environ["SPLITS"].split()] + [-1]


def reps_for(M, N, K, B):
    return max(10, min(300, int(0.12 / (2.0 * M * N * K * B / 1.0e14 + 5e-6))))


def ours(M, N, K, B, tr, knob):
    a = device_random(wg, gpu, (K, M, B) if tr else (M, K, B), np.float32, 1)
    b = device_random(wg, gpu, (K, N, B), np.float32, 2)
    c = wg.TensorBuilder.tensor((M, N, B), S.STORAGE).build(dev, np.float32)
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    split = 0
    if isinstance(knob, tuple):
        knob, split = knob
    old = gpu.set_tuning("f32_mid", knob)
    old_split = gpu.set_tuning("f32_mid_split", split)

    def run(n):
        enc = dev.create_command_encoder()
        p = enc.compute_pass("t", None)
        for _ in range(n):
            gemm.dispatch_generic(dev, shapes, p, c, a, b, variant)
        p.end()
        gpu.queue().submit([enc.finish()])
        gpu.sync()
    try:
        run(3)
        n = reps_for(M, N, K, B)
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            run(n)
            best = min(best, (time.perf_counter() - t0) / n)
    finally:
        gpu.set_tuning("f32_mid", old)
        gpu.set_tuning("f32_mid_split", old_split)
    return best


def vendor(M, N, K, B, tr):
    bt = torch.rand(B, N, K, device="cuda") * 2 - 1
    at = (torch.rand(B, M, K, device="cuda") * 2 - 1).transpose(1, 2) if tr else torch.rand(B, K, M, device="cuda") * 2 - 1
    ct = torch.empty(B, N, M, device="cuda")
    for _ in range(3):
        torch.matmul(bt, at, out=ct)
    torch.cuda.synchronize()
    n = reps_for(M, N, K, B)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(n):
            torch.matmul(bt, at, out=ct)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best


if os.environ.get("SPLITS"):
    print("# us per call: round-3 paths (knob 0) | 64x64 tile with K cut in " + " / ".join(os.environ["SPLITS"].split()) + " | launcher's choice | vendor")
print("# us per call: gemm_f32.hip plan | 128x128 | 128x64 | 64x128 | 64x64 | 64x32 | 32x64 | 96x96 | 96x64 | launcher's choice | vendor ; choice/vendor")
for (M, N, K, B) in SHAPES:
    for tr in (False, True):
        ts = [ours(M, N, K, B, tr, k) * 1e6 for k in KNOBS]
        v = vendor(M, N, K, B, tr) * 1e6
        fl = 2.0 * M * N * K * B
        flag = "  <-- behind" if ts[-1] > 1.10 * v else ""
        print(f"f32 {'gemm_tr' if tr else 'gemm   '} {M}x{N}x{K}x{B}: " + " | ".join(f"{t:7.1f}" for t in ts) + f" | vendor {v:7.1f} ; {ts[-1] / v:5.2f}  ({fl / ts[-1] / 1e6:6.1f} TF){flag}", flush=True)
