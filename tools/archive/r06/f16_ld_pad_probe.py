"""f16 Gemm / GemmTr with padded leading dimensions over the tile families (knob f16_tile): where does a row pitch that is not a multiple of 64 / 128 bytes cost?"""
import os, sys
sys.argv = [sys.argv[0], "none"]
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, root + "/tools")
import numpy as np
src = open(root + "/tools/cliff_sweep.py").read().split("for name in (sys.argv[1:]")[0]
ns = {"__file__": root + "/tools/cliff_sweep.py", "__name__": "cs"}
exec(compile(src, "cs", "exec"), ns)
gpu, gemm_case = ns["gpu"], ns["gemm_case"]
for tr in (False, True):
    for (M, N, K) in [(2048, 2048, 2048), (4096, 4096, 4096), (8192, 8192, 1024), (4096, 2048, 2048), (3072, 3072, 3072), (8192, 8192, 8192)]:
        for pad in [(0, 0, 0), (8, 8, 8), (32, 32, 32), (64, 64, 64), (8, 0, 0), (0, 8, 0), (0, 0, 8)]:
            row = []
            for knob in (0, 128, 256, 256128):
                old = gpu.set_tuning("f16_tile", knob)
                try:
                    row.append(f"{knob}:{gemm_case(np.float16, tr, M, N, K, pad):7.1f}")
                except Exception as e:
                    row.append(f"{knob}:   err")
                gpu.set_tuning("f16_tile", old)
            print(f"f16 {'tr' if tr else 'nn'} {M}x{N}x{K} pad {pad}: " + "  ".join(row), flush=True)
