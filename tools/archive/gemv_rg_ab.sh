# A/B of gemv_n_kernel's row groups per lane (WG_GEMV_RG): bench workloads + the N sweep, per variant library (tools/build_variant.sh rgN gemv.hip "-DWG_GEMV_RG=N")
cd $GRAFT_REPO_ROOT
for wl in gemv_f32_4096x65536 gemv_f16_4096x65536; do
for v in "" _rg4n2 _rg4n1 _rg8n1 _rg2n4 ""; do
  r=$(WG_BENCH_NO_CEILING=1 WGEBRA_HIP_LIB=$GRAFT_REPO_ROOT/wgmath_amd/libwgebra_hip$v.so timeout 200 python bench.py --workload $wl --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline']['frac'])")
  echo "$wl lib$v: $r"
done; done
