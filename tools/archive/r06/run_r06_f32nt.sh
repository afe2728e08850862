#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "row_major" 2>&1 | tail -5
one() { WG_BENCH_NO_CHECK=${NOCHECK:-1} python bench.py --steps 100 --warmup 10 --workload $1 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['roofline'].get('clock_ghz_measured'))"; }
{
NOCHECK=0 one gemmtr_rm_f32_4096 | sed 's/^/checked run: /'
for round in 1 2 3; do
  echo "gemmtr_rm_f32_4096 native $(one gemmtr_rm_f32_4096)"
  echo "gemmtr_rm_f32_4096 copy   $(WG_RM_TR_NATIVE=0 one gemmtr_rm_f32_4096)"
  echo "gemmtr_f32_4096 (col-major TN) $(one gemmtr_f32_4096)"
  echo "gemm_f32_4096 (col-major NN) $(one gemm_f32_4096)"
done
} > gpurun_out/r06_f32_nt_ab.txt 2>&1
cat gpurun_out/r06_f32_nt_ab.txt
python tools/gemm_sweep.py rm f32 4096x4096x4096 8192x8192x1024 4100x4104x264 8192x2048x2048 2>&1 | grep -v "amdgpu\|^RCCL"
