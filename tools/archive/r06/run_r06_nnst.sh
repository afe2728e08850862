#!/bin/bash
# round 6: Gemm on the staged pipeline (one barrier per stage) now that its fragment path is swap-free: parity of the variant, then A/B, interleaved
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
WGEBRA_HIP_LIB=$PWD/wgmath_amd/libwgebra_hip_nnst.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "f16 and (bit_identical or families or remainder or ragged or split or tail)" 2>&1 | tail -3
{
for round in 1 2 3; do
  for wl in gemm_f16_32768 gemm_f16_16384x16384x8192 gemm_f16_ts_131072x1024x8192; do
    st=60; [ $wl = gemm_f16_32768 ] && st=20
    STEPS=$st bash tools/ab2.sh $wl libwgebra_hip.so libwgebra_hip_nnst.so | sed "s/^/$wl /"
  done
done
} > gpurun_out/r06_nnst_ab.txt 2>&1
cat gpurun_out/r06_nnst_ab.txt
