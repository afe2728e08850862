cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks3
WG_BENCH_NO_CEILING=1 WG_BENCH_NO_CHECK=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks3 -o k -- python3 $GRAFT_REPO_ROOT/bench.py --workload gemv_f32_4096x65536_rhs8 --no-secondary --no-cpu-baseline --steps 50 --warmup 5 > /tmp/ks3.json 2>/dev/null
f=$(find /tmp/ks3 -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-260; tail -c 600 /tmp/ks3.json
