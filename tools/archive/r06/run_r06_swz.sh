#!/bin/bash
# round 6: the few-column f32 kernel's permuted [32 k][32 m] stage image (no LDS bank conflicts): parity, then A/B against -DWG_SKINNY_SWZ=0, interleaved
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_tests.py -q -x -m gpu -k "few or skinny or gemm_shapes or gemm_golden or reference or multi_rhs or panels" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_cpp_facade.py -q -x -m gpu 2>&1 | tail -3
{
for round in 1 2 3; do
  for lib in libwgebra_hip.so libwgebra_hip_swz0.so; do
    echo "fewcols_32000x16x4096 $lib $(WGEBRA_HIP_LIB=$PWD/wgmath_amd/$lib WG_BENCH_NO_CHECK=1 python bench.py --steps 2000 --warmup 50 --workload gemm_f32_fewcols_32000x16x4096 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])")"
  done
done
for lib in libwgebra_hip.so libwgebra_hip_swz0.so; do
  echo "== $lib"; WGEBRA_HIP_LIB=$PWD/wgmath_amd/$lib python tools/gemm_sweep.py f32 32000x16x4096 32000x8x4096 65536x4x4096 16384x32x4096 8192x64x8192 4096x16x4096 16x4096x4096 64x4096x4096 2>&1 | grep -v "^RCCL\|gemm_tr"
done
} > gpurun_out/r06_skinny_swz_ab.txt 2>&1
cat gpurun_out/r06_skinny_swz_ab.txt
