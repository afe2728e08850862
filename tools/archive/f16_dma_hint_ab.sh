# A/B of a cache hint on the 256 x 256 f16 kernels' operand pieces (WG_F16_DMA_HINT; tools/build_variant.sh dhN gemm_f16.hip "-DWG_F16_DMA_HINT=N")
cd $GRAFT_REPO_ROOT
for wl in gemm_f16_32768 gemm_f16_8192 gemm_f16_ts_131072x1024x8192; do
for v in "" _dh1 _dh2 _dh3 ""; do
  st=40; [ $wl = gemm_f16_32768 ] && st=8
  r=$(WG_BENCH_NO_CEILING=1 WG_BENCH_NO_CHECK=1 WGEBRA_HIP_LIB=$GRAFT_REPO_ROOT/wgmath_amd/libwgebra_hip$v.so timeout 300 python bench.py --workload $wl --steps $st --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline']['frac'], d['roofline'].get('clock_ghz_measured'))")
  echo "$wl lib$v: $r"
done; done
