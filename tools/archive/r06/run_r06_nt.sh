#!/bin/bash
# round 6: the row-major GemmTr kernel (gemm_f16_nt.hip): parity, then against the transposed-copy path and the column-major kernels, interleaved
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "row_major" 2>&1 | tail -5
timeout 300 python -m pytest tests/test_gpu_dist2.py -q -x -m gpu -k "back_to_back" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_cpp_facade.py -q -x -m gpu -k "comm_tests_on_gpu" 2>&1 | tail -3
tests/cpp/_build/comm_tests 2>&1 | grep comm_tests
one() { WG_BENCH_NO_CHECK=${NOCHECK:-1} python bench.py --steps 200 --warmup 20 --workload $1 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['roofline'].get('clock_ghz_measured'))"; }
{
NOCHECK=0 one gemmtr_rm_f16_8192 | sed 's/^/checked run: /'
for round in 1 2 3; do
  echo "gemmtr_rm_f16_8192 native $(one gemmtr_rm_f16_8192)"
  echo "gemmtr_rm_f16_8192 copy   $(WG_RM_TR_NATIVE=0 one gemmtr_rm_f16_8192)"
  echo "gemmtr_f16_8192 (col-major TN) $(one gemmtr_f16_8192)"
  echo "gemm_f16_8192 (col-major NN) $(one gemm_f16_8192)"
  echo "gemm_rm_f16_8192 $(one gemm_rm_f16_8192)"
done
echo "gemmtr_rm_f32_4096 $(one gemmtr_rm_f32_4096)"
echo "gemmtr_f32_4096 $(one gemmtr_f32_4096)"
} > gpurun_out/r06_nt_ab.txt 2>&1
cat gpurun_out/r06_nt_ab.txt
