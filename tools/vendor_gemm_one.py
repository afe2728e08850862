import sys, torch
n = int(sys.argv[1]); layout = sys.argv[2]
import os
z = 0.0 if os.environ.get("WG_BENCH_VALUES") == "zero" else 1.0
a = ((torch.rand(n, n, device="cuda") * 2 - 1) * z).half(); b = ((torch.rand(n, n, device="cuda") * 2 - 1) * z).half()
if layout == "nt": b = b.t().contiguous().t()
if layout == "tn": a = a.t().contiguous().t()
c = torch.empty(n, n, device="cuda", dtype=torch.float16)
for _ in range(20): torch.matmul(a, b, out=c)
torch.cuda.synchronize()
