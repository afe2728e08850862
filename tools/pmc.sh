#!/bin/bash
# Machine-readable PMC summary of the SHIPPED kernels on the BASELINE configs -> gpurun_out/pmc_$R/${R}_pmc.csv (copy to profiles/); R = round tag (default r03).
# One rocprofv3 run per workload; tools/pmc_passes.txt holds the counter passes (separate --pmc passes, --kernel-trace only: no
# other trace domain next to the counters). Usage (GPU box, repo root): bash tools/pmc.sh [workload ...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
R=${R:-r06}
OUT=$ROOT/gpurun_out/pmc_$R
mkdir -p $OUT
WLS=${@:-gemm_f16_8192 gemmtr_f16_8192 gemm_f16_32768 gemmtr_f16_32768 gemmtr_rm_f16_8192 gemm_f32_4096 gemm_f16_2048 gemm_f32_2048 gemm_f16_ts_131072x1024x8192 gemm_f16_8192x8192x1024 gemmtr_f16_8192x8192x1024 gemm_f32_fewcols_32000x16x4096 gemv_f32_4096x65536 gemvtr_f32_65536x4096 gemv_f16_4096x65536 gemvtr_f16_65536x4096 reduce_f32_4096x65536 op_assign_f32_256M}
cd /tmp && export TMPDIR=/tmp
for wl in $WLS; do
  rm -rf $OUT/$wl
  steps=6; [ $wl = gemm_f16_32768 ] && steps=3; [ $wl = gemmtr_f16_32768 ] && steps=3
  WG_BENCH_NO_CHECK=1 WG_BENCH_NO_CEILING=1 rocprofv3 -i $ROOT/tools/pmc_passes.txt --kernel-trace --output-format csv -d $OUT/$wl -o p -- python3 $ROOT/bench.py --steps $steps --warmup 2 --workload $wl --no-secondary --no-cpu-baseline > $OUT/$wl.json 2> $OUT/$wl.log
done
cd $ROOT
python3 tools/pmc_post.py $OUT $R $WLS
