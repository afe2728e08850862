#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_tests.py -q -x -m gpu -k "gemm and not f16" 2>&1 | tail -3
{
for lib in libwgebra_hip.so libwgebra_hip_noflat.so; do echo "== $lib"; WGEBRA_HIP_LIB=$PWD/wgmath_amd/$lib python tools/gemm_sweep.py f32 4096x7168x2048x3 1280x2816x6144x8 1024x11008x4096x8 12288x2048x16384x3 3072x1408x16384x8 4096x4096x4096x3 2048x2048x2048x8 1536x1536x1536x5 4100x4104x264x3 2>&1 | grep "float32"; done
} > gpurun_out/r06_f32_flat_batch_ab.txt 2>&1
cat gpurun_out/r06_f32_flat_batch_ab.txt
