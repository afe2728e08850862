set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -k "mid" -m gpu -q 2>&1 | tail -8 > gpurun_out/r04_tests5.log
tail -5 gpurun_out/r04_tests5.log
timeout 1500 python tools/f32_mid_sweep.py 512x512x512 1024x1024x1024 1000x1000x1000 1536x1536x1536 2048x2048x2048 2560x2560x2560 3072x3072x3072 1024x1024x4096 4096x1024x1024 2048x2048x512 512x512x512x8 256x256x256x64 128x128x128x256 1024x1024x1024x4 1024x1024x256 1024x1024x512 1024x1024x2048 > gpurun_out/r04_f32_mid_sweep3.txt 2>&1
cat gpurun_out/r04_f32_mid_sweep3.txt
