#!/bin/bash
# round 6 experiment: staggered starts of the continuous walk (do the CUs' synchronized store bursts at tile boundaries cost what the walk's 5.7 us per round says?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { WG_BENCH_NO_CHECK=1 python bench.py --steps 300 --warmup 30 --workload $1 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['roofline'].get('clock_ghz_measured'))"; }
{
for round in 1 2; do
for wl in gemm_f16_8192x8192x1024 gemmtr_f16_8192x8192x1024 gemm_f16_8192x8192x512 gemm_f16_8192x8192x2048 gemm_f16_8192; do
  for st in "" "50,4" "100,4" "150,4" "250,4" "100,8" "200,2" "60,16"; do
    echo "$wl stagger=[$st] $(WG_F16_STAGGER=$st one $wl)"
  done
done
done
} > gpurun_out/r06_stagger_ab.txt 2>&1
cat gpurun_out/r06_stagger_ab.txt
