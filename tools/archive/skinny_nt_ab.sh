cd $GRAFT_REPO_ROOT
for wl in gemm_f32_fewcols_32000x16x4096 gemv_f32_4096x65536_rhs8; do
for v in "" _ant1 _ant2 _ant3 ""; do
  r=$(WG_BENCH_NO_CEILING=1 WGEBRA_HIP_LIB=$GRAFT_REPO_ROOT/wgmath_amd/libwgebra_hip$v.so timeout 200 python bench.py --workload $wl --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline']['frac'])")
  echo "$wl lib$v: $r"
done; done
