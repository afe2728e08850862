#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "row_major" 2>&1 | tail -5
{
for nat in -1 0; do echo "== WG_RM_TR_NATIVE=$nat"; WG_RM_TR_NATIVE=$nat python tools/gemm_sweep.py rm f16 2048x2048x2048 1024x1024x1024 4096x4096x1024 2048x2048x2048x8 1024x4096x2048 3072x3072x3072 1536x1536x4096 512x4096x4096 2>&1 | grep "gemm_tr"; done
} > gpurun_out/r06_row_major_midsize_ab.txt 2>&1
cat gpurun_out/r06_row_major_midsize_ab.txt
