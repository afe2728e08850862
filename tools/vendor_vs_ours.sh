#!/bin/bash
# Same box, same moment: hipBLASLt (through torch.matmul; tools/vendor_gemm_probe.py method) and this library on f16 8192^3 / 32768^3.
python3 - <<'PY'
import time, torch
def run(n, layout, seconds=2.0):
    a = (torch.rand(n, n, device="cuda", dtype=torch.float32) * 2 - 1).to(torch.float16)
    b = (torch.rand(n, n, device="cuda", dtype=torch.float32) * 2 - 1).to(torch.float16)
    if layout == "nt": b = b.t().contiguous().t()
    c = torch.empty(n, n, device="cuda", dtype=torch.float16)
    for _ in range(3): torch.matmul(a, b, out=c)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); torch.matmul(a, b, out=c); torch.cuda.synchronize(); one = time.perf_counter() - t0
    iters = max(5, int(seconds / one))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): torch.matmul(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    return 2.0 * n ** 3 / (e0.elapsed_time(e1) / iters) / 1e9
for n in (8192, 16384):
    for layout in ("nn", "nt"):
        print(f"vendor f16 {n}^3 {layout}: {run(n, layout):8.1f} TFLOP/s", flush=True)
PY
for w in gemm_f16_8192 gemmtr_f16_8192 gemm_f16_32768; do
  st=100; [ $w = gemm_f16_32768 ] && st=10
  echo "ours $w: $(WG_BENCH_NO_CHECK=1 python3 bench.py --steps $st --warmup 10 --workload $w --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")"
done
