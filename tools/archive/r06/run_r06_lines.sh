#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
WGEBRA_HIP_LIB=$PWD/wgmath_amd/libwgebra_hip_lines.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "continuous or cont" 2>&1 | tail -2
{
for round in 1 2 3; do
  for wl in gemm_f16_8192x8192x1024 gemmtr_f16_8192x8192x1024 gemm_f16_8192x8192x512 gemm_f16_8192x8192x2048 gemm_f16_8192; do
    STEPS=300 bash tools/ab2.sh $wl libwgebra_hip.so libwgebra_hip_lines.so | grep rand | sed "s/^/$wl /"
  done
done
} > gpurun_out/r06_cont_store_lines_ab.txt 2>&1
cat gpurun_out/r06_cont_store_lines_ab.txt
