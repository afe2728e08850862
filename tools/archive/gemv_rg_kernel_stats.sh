cd /tmp && export TMPDIR=/tmp
for v in "" _rg1; do
  rm -rf /tmp/ks$v
  WG_BENCH_NO_CEILING=1 WG_BENCH_NO_CHECK=1 WGEBRA_HIP_LIB=$GRAFT_REPO_ROOT/wgmath_amd/libwgebra_hip$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks$v -o k -- python3 $GRAFT_REPO_ROOT/bench.py --workload gemv_f32_4096x65536 --no-secondary --no-cpu-baseline --steps 50 --warmup 5 > /dev/null 2>&1
  echo "== lib$v"; f=$(find /tmp/ks$v -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-200
done
