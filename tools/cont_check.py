#!/usr/bin/env python3
"""f16 Gemm / GemmTr, continuous tile walk (WG_TUNE_F16_CONT = 1) against the per-tile launch (0) and the default rule (-1): same bits, and the GPU time of each, per shape.

    [CONT_VARIANT=gemm|gemmtr] [CONT_TILE=256] python tools/cont_check.py [MxNxK ...]      (GPU box; exit code 1 when any shape differs or leaves a NaN)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import wgmath_amd as wg  # noqa: E402

SHAPES = [(8192, 8192, 1024), (8192, 8192, 256), (8192, 8192, 320), (5120, 5120, 640), (2560, 8192, 256), (8192, 8192, 4096), (4352, 4096, 448), (8192, 8192, 640),
          (8192, 8192, 768), (8192, 8192, 1280), (8192, 8192, 2048), (16384, 16384, 1024)]
S_STORAGE = 128 | 4 | 8


def main():
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or SHAPES
    gpu = wg.GpuInstance.new(0)
    dev, vs = gpu.device(), wg.ViewShapeBuffers()
    gemm = wg.Gemm.from_device(dev)
    if os.environ.get("CONT_TILE"):  # e.g. 256: the 256 x 256 family also where the 256 x 128 tile would take the shape (short K)
        gpu.set_tuning("f16_tile", int(os.environ["CONT_TILE"]))
    tr = os.environ.get("CONT_VARIANT", "gemmtr") == "gemmtr"
    variant = wg.GemmVariant.GemmTr if tr else wg.GemmVariant.Gemm
    rng = np.random.default_rng(5)
    bad = 0
    for M, N, K in shapes:
        a = (rng.random(M * K, dtype=np.float32) * 2 - 1).astype(np.float16)
        b = (rng.random(K * N, dtype=np.float32) * 2 - 1).astype(np.float16)
        m1 = wg.TensorBuilder.tensor((K, M, 1) if tr else (M, K, 1), S_STORAGE).build_init(dev, a, np.float16)
        m2 = wg.TensorBuilder.tensor((K, N, 1), S_STORAGE).build_init(dev, b, np.float16)
        res, us = {}, {}
        for cont in (0, 1, -1):
            gpu.set_tuning("f16_cont", cont)
            out = wg.TensorBuilder.tensor((M, N, 1), S_STORAGE).build_init(dev, np.full(M * N, np.nan, np.float16), np.float16)

            def run(n):
                enc = dev.create_command_encoder()
                p = enc.compute_pass("t", None)
                for _ in range(n):
                    gemm.dispatch_generic(dev, vs, p, out, m1, m2, variant)
                p.end()
                gpu.queue().submit([enc.finish()])
                gpu.sync()
            run(1)
            res[cont] = out.read(dev).view(np.uint16).copy()
            run(3)
            n = max(10, min(200, int(0.1 / (2.0 * M * N * K / 1.2e15 + 5e-6))))
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                run(n)
                best = min(best, (time.perf_counter() - t0) / n)
            us[cont] = best * 1e6
        same = np.array_equal(res[0], res[1])
        nan = int(np.isnan(res[1].view(np.float16)).sum())
        bad += (not same) or nan > 0
        nd = int((res[0] != res[1]).sum())
        same = same and np.array_equal(res[0], res[-1])
        bad += not same
        print(f"{'gemm_tr' if tr else 'gemm   '} {M}x{N}x{K}: per-tile {us[0]:8.1f} us  continuous {us[1]:8.1f} us  ({2.0 * M * N * K / us[1] / 1e6:7.1f} TF)  by default {us[-1]:8.1f} us  same bits {same} (differing {nd}, nan {nan})",
              flush=True)
    gpu.set_tuning("f16_cont", -1)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
