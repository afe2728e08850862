set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/f32_mid_sweep.py > gpurun_out/r04_f32_mid_sweep.txt 2>&1
grep -c behind gpurun_out/r04_f32_mid_sweep.txt; grep behind gpurun_out/r04_f32_mid_sweep.txt | cut -c1-200
timeout 2400 python tools/gemm_sweep.py > gpurun_out/r04_gemm_sweep_full.txt 2>&1
grep -c behind gpurun_out/r04_gemm_sweep_full.txt; grep behind gpurun_out/r04_gemm_sweep_full.txt
