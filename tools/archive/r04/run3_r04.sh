set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -k "mid" -m gpu -q 2>&1 | tail -30 > gpurun_out/r04_tests3.log
tail -12 gpurun_out/r04_tests3.log
timeout 1500 python tools/f32_mid_sweep.py > gpurun_out/r04_f32_mid_sweep.txt 2>&1
cat gpurun_out/r04_f32_mid_sweep.txt
