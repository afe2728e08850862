#!/bin/bash
# round 6: the swap-free Gemm A path -- parity first (bit-identity to the other kernel families and to round 5's form), then A/B against -DWG_NN_NOSWAP=0, interleaved
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ulp.py -q -x -m gpu -k "f16" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "f16 or config3 or config5 or c3 or c5" 2>&1 | tail -5
{
for round in 1 2 3; do
  for wl in gemm_f16_8192 gemm_f16_32768 gemm_f16_8192x8192x1024 gemm_f16_ts_131072x1024x8192; do
    st=200; [ $wl = gemm_f16_32768 ] && st=20
    STEPS=$st bash tools/ab2.sh $wl libwgebra_hip.so libwgebra_hip_swap.so | sed "s/^/$wl /"
  done
  STEPS=200 bash tools/ab2.sh gemmtr_f16_8192 libwgebra_hip.so | sed "s/^/gemmtr_f16_8192 /"
done
} > gpurun_out/r06_noswap_ab.txt 2>&1
cat gpurun_out/r06_noswap_ab.txt
