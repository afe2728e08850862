set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/f32_mid_sweep.py > gpurun_out/r04_f32_mid_sweep4.txt 2>&1
cat gpurun_out/r04_f32_mid_sweep4.txt | cut -c1-200
timeout 1800 python tools/gemm_sweep.py f32 > gpurun_out/r04_gemm_sweep_f32.txt 2>&1
grep -c behind gpurun_out/r04_gemm_sweep_f32.txt; grep behind gpurun_out/r04_gemm_sweep_f32.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_tests.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r04_tests6.log
cat gpurun_out/r04_tests6.log
