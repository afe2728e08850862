#!/bin/bash
# round 6: vendor library beside ours on the same box and moment (hipBLASLt through torch.matmul), the fixed shape sweeps incl. the row-major surface, the rank emulation
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
bash tools/vendor_vs_ours.sh > gpurun_out/r06/r06_vendor_vs_ours.txt 2>&1
python tools/gemm_sweep.py rm f16 8192x8192x8192 4096x4096x4096 2048x2048x2048 8192x8192x1024 4096x11008x4096 16384x16384x2048 8192x4096x4096 4104x4104x4104 1024x1024x1024 2048x2048x2048x8 > gpurun_out/r06/r06_gemm_sweep_row_major.txt 2>&1
python tools/gemm_sweep.py rm f32 4096x4096x4096 2048x2048x2048 8192x8192x1024 >> gpurun_out/r06/r06_gemm_sweep_row_major.txt 2>&1
python tools/gemm_sweep.py > gpurun_out/r06/r06_gemm_sweep_full.txt 2>&1
python tools/misc_sweep.py > gpurun_out/r06/r06_misc_sweep.txt 2>&1
export STEPS=6
for P in 2 4 8; do python tools/rank_emulation.py $P > gpurun_out/r06/r06_rank_emulation_p$P.json 2> gpurun_out/r06/rank_emulation_p$P.log; done
grep -v amdgpu gpurun_out/r06/r06_vendor_vs_ours.txt; grep -c behind gpurun_out/r06/r06_gemm_sweep_full.txt gpurun_out/r06/r06_misc_sweep.txt gpurun_out/r06/r06_gemm_sweep_row_major.txt; grep -v "amdgpu\|^RCCL" gpurun_out/r06/r06_gemm_sweep_row_major.txt
