cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
SHAPES="8192x8192x128 8192x8192x256 8192x8192x384 8192x8192x512 8192x8192x768 8192x8192x1024 8192x8192x1280 8192x8192x2048 4096x4096x256 4096x4096x512 4096x4096x1024 16384x16384x256 16384x16384x512 6144x6144x512 4096x8192x512 5120x5120x640 32768x4096x512 4096x32768x256"
{ echo "# tools/gemm_sweep.py f16 <short-K shapes>: launcher's choice (final code)"; timeout 500 python tools/gemm_sweep.py f16 $SHAPES 2>&1 | grep float16
  echo "# WG_F16_TILE=256: the 256 x 256 kernel forced (the plan before the 256 x 128 tile)"; WG_F16_TILE=256 timeout 500 python tools/gemm_sweep.py f16 $SHAPES 2>&1 | grep float16
  echo "# WG_F16_TILE=256128: the 256 x 128 tile forced"; WG_F16_TILE=256128 timeout 500 python tools/gemm_sweep.py f16 $SHAPES 2>&1 | grep float16; } > gpurun_out/r04_f16_short_k_sweep.txt
grep -c behind gpurun_out/r04_f16_short_k_sweep.txt
