#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { WGEBRA_HIP_LIB=$PWD/wgmath_amd/libwgebra_hip_permst.so WG_BENCH_NO_CHECK=1 python bench.py --steps 300 --warmup 30 --workload $1 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])"; }
{
for round in 1 2; do
for wl in gemm_f16_8192x8192x1024 gemmtr_f16_8192x8192x1024 gemm_f16_8192x8192x512 gemm_f16_8192x8192x2048; do
  for st in "" "30,4" "60,4" "100,4" "40,8" "100,2" "20,16"; do
    echo "$wl perm + stagger=[$st] $(WG_F16_STAGGER=$st one $wl)"
  done
done
done
} > gpurun_out/r06_cont_perm_stagger_ab.txt 2>&1
cat gpurun_out/r06_cont_perm_stagger_ab.txt
