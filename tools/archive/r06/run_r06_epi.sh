#!/bin/bash
# round 6: the continuous walk's lighter epilogue (selects on packed halves, no multiply by alpha == 1) against the build before it (libwgebra_hip_nnst.so: same cont kernel as before), interleaved
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "continuous or cont or f16_shapes or alpha" 2>&1 | tail -3
{
for round in 1 2 3; do
  for wl in gemm_f16_8192x8192x1024 gemmtr_f16_8192x8192x1024 gemm_f16_8192x8192x512 gemm_f16_8192 gemmtr_f16_8192; do
    STEPS=300 bash tools/ab2.sh $wl libwgebra_hip.so libwgebra_hip_nnst.so | grep rand | sed "s/^/$wl /"
  done
done
} > gpurun_out/r06_cont_epilogue_ab.txt 2>&1
cat gpurun_out/r06_cont_epilogue_ab.txt
