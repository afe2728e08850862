import os, sys
sys.argv = [sys.argv[0], "none"]
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + ""); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools")
os.environ["CLIFF_ONLY"] = "none"
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("cs", os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools/cliff_sweep.py")
src = open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools/cliff_sweep.py").read().split("for name in (sys.argv[1:]")[0]
ns = {"__file__": os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools/cliff_sweep.py", "__name__": "cs"}
exec(compile(src, "cs", "exec"), ns)
gpu, gemm_case = ns["gpu"], ns["gemm_case"]
for tr in (False, True):
    for (M, N, K) in [(2048, 2048, 2048), (2048, 2048, 1024), (2048, 2048, 4096), (3072, 3072, 1024), (1024, 4096, 2048), (4096, 1024, 1024), (1920, 1920, 2048)]:
        for pad in [(0, 0, 0), (8, 8, 8), (64, 64, 64), (8, 0, 0), (0, 8, 0)]:
            row = []
            for knob in (-1, 128064, 64064):
                old = gpu.set_tuning("f32_mid", knob)
                try:
                    row.append(f"{knob}:{gemm_case(np.float32, tr, M, N, K, pad):7.1f}")
                except Exception as e:
                    row.append(f"{knob}:   err")
                gpu.set_tuning("f32_mid", old)
            print(f"f32 {'tr' if tr else 'nn'} {M}x{N}x{K} pad {pad}: " + "  ".join(row), flush=True)
