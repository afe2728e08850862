set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -k "mid" -m gpu -q 2>&1 | tail -8 > gpurun_out/r04_tests9.log
tail -5 gpurun_out/r04_tests9.log
SPLITS="0 2 4 8 16" timeout 1200 python tools/f32_mid_sweep.py 64x4096x4096 4096x64x4096 128x4096x4096 4096x128x4096 64x8192x8192 256x256x8192 512x512x4096 256x256x32768 64x11008x4096 1024x1024x32768 > gpurun_out/r04_f32_split_sweep.txt 2>&1
cat gpurun_out/r04_f32_split_sweep.txt | cut -c1-220
