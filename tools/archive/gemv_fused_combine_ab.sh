# A/B of Gemv's fused combine (WG_GEMV_FUSED_COMBINE): bench workloads per variant library (tools/build_variant.sh nofuse gemv.hip "-DWG_GEMV_FUSED_COMBINE=0")
cd $GRAFT_REPO_ROOT
for wl in gemv_f32_4096x65536 gemv_f32_4096x65536_rhs8 gemv_f16_4096x65536; do
for v in "" _nofuse "" _nofuse; do
  r=$(WG_BENCH_NO_CEILING=1 WGEBRA_HIP_LIB=$GRAFT_REPO_ROOT/wgmath_amd/libwgebra_hip$v.so timeout 200 python bench.py --workload $wl --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline']['frac'], d.get('checks'))")
  echo "$wl lib$v: $r"
done; done
