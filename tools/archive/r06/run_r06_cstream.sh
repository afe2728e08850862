#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for round in 1 2; do
  for wl in gemm_f16_8192x8192x1024 gemmtr_f16_8192x8192x1024 gemm_f16_8192x8192x512 gemm_f16_8192x8192x2048 gemm_f16_8192; do
    STEPS=300 bash tools/ab2.sh $wl libwgebra_hip_cstream0.so libwgebra_hip_cstream1.so | grep rand | sed "s/^/$wl /"
  done
done
} > gpurun_out/r06_cont_cstream_ab.txt 2>&1
cat gpurun_out/r06_cont_cstream_ab.txt
