# A/B of OpAssign on a contiguous slab per workgroup (OPA_MODE 3) against the flat grid of one float4 per lane (shipped): bench workload per variant library
cd $GRAFT_REPO_ROOT
for v in "" _opa_u4_w8 _opa_u8_w8 _opa_u4_w4 _opa_u8_w4 _opa_u2_w16 _opa_u4_w2 ""; do
  r=$(WG_BENCH_NO_CEILING=1 WGEBRA_HIP_LIB=$GRAFT_REPO_ROOT/wgmath_amd/libwgebra_hip$v.so timeout 200 python bench.py --workload op_assign_f32_256M --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline']['frac'])")
  echo "op_assign_f32_256M lib$v: $r"
done
