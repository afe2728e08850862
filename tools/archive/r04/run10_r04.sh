set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_tests.py tests/test_gpu_ulp.py -m gpu -q 2>&1 | tail -8 > gpurun_out/r04_tests10.log
tail -5 gpurun_out/r04_tests10.log
SPLITS="0 4" timeout 1200 python tools/f32_mid_sweep.py 64x4096x4096 4096x64x4096 128x4096x4096 64x8192x8192 256x256x8192 512x512x4096 256x256x32768 64x11008x4096 4096x16x4096 32000x16x4096 16x4096x4096 48x4096x4096 > gpurun_out/r04_f32_split_sweep2.txt 2>&1
cat gpurun_out/r04_f32_split_sweep2.txt | cut -c1-200
timeout 2400 python tools/gemm_sweep.py f32 > gpurun_out/r04_gemm_sweep_f32b.txt 2>&1
grep -c behind gpurun_out/r04_gemm_sweep_f32b.txt; grep behind gpurun_out/r04_gemm_sweep_f32b.txt
