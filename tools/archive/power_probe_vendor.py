#!/usr/bin/env python3
"""Package power / sclk while the vendor GEMM (torch.matmul -> hipBLASLt) runs for a few seconds: tools/power_probe_vendor.py <layout nn|nt|tn>.
The companion of tools/power_probe.sh (same rocm-smi sampling) for the energy-per-flop comparison of profiles/r02_evidence.md."""
import subprocess, sys, threading, time, torch
layout = sys.argv[1] if len(sys.argv) > 1 else "nt"
n = 8192
a = (torch.rand(n, n, device="cuda") * 2 - 1).half(); b = (torch.rand(n, n, device="cuda") * 2 - 1).half()
if layout == "nt": b = b.t().contiguous().t()
if layout == "tn": a = a.t().contiguous().t()
c = torch.empty(n, n, device="cuda", dtype=torch.float16)
for _ in range(50): torch.matmul(a, b, out=c)
torch.cuda.synchronize()
samples = []
def sample():
    time.sleep(2.0)
    for _ in range(6):
        out = subprocess.run("rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Package Power|sclk'", shell=True, capture_output=True, text=True).stdout
        samples.append(" | ".join(l.split(":", 1)[-1].strip() for l in out.strip().splitlines()))
        time.sleep(0.5)
t = threading.Thread(target=sample); t.start()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 9000
e0.record()
for _ in range(iters): torch.matmul(a, b, out=c)
e1.record(); torch.cuda.synchronize(); t.join()
ms = e0.elapsed_time(e1) / iters
for s in samples: print(s)
print(f"vendor {layout}: {2.0 * n ** 3 / ms / 1e9:.1f} TFLOP/s, {ms * 1e3:.1f} us")
