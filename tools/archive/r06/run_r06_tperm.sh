#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -m gpu -k "f16" 2>&1 | tail -3
{
for round in 1 2 3; do
  for wl in gemm_f16_32768 gemmtr_f16_32768 gemm_f16_16384x16384x8192 gemm_f16_ts_131072x1024x8192; do
    st=60; [ $wl = gemm_f16_32768 ] && st=20; [ $wl = gemmtr_f16_32768 ] && st=20
    STEPS=$st bash tools/ab2.sh $wl libwgebra_hip.so libwgebra_hip_tperm0.so | grep rand | sed "s/^/$wl /"
  done
done
} > gpurun_out/r06_tile_store_perm_ab.txt 2>&1
cat gpurun_out/r06_tile_store_perm_ab.txt
