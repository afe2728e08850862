set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -k "mid" -m gpu -q 2>&1 | tail -8 > gpurun_out/r04_tests7.log
tail -5 gpurun_out/r04_tests7.log
timeout 1500 python tools/f32_mid_sweep.py 768x768x768 1536x1536x1536 3072x3072x3072 1152x1152x1152 1920x1920x1920 2304x2304x2304 1536x1536x512 1536x3072x1024 4096x4096x4096 5120x5120x2048 6144x6144x6144 8192x8192x8192 4096x4096x11008 1024x8192x8192 8200x8200x8200 > gpurun_out/r04_f32_mid_sweep5.txt 2>&1
cat gpurun_out/r04_f32_mid_sweep5.txt | cut -c1-230
