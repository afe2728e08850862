#!/bin/bash
# Machine-readable PMC summary of the SHIPPED kernels on the BASELINE configs -> gpurun_out/pmc_r02/r02_pmc.csv (copy to profiles/).
# One rocprofv3 run per workload; tools/pmc_r02.txt holds the counter passes (separate --pmc passes, --kernel-trace only: no
# other trace domain next to the counters). Usage (GPU box, repo root): bash tools/pmc_r02.sh [workload ...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_r02
mkdir -p $OUT
WLS=${@:-gemm_f16_8192 gemmtr_f16_8192 gemm_f16_32768 gemm_f32_4096 gemm_f16_2048 gemm_f16_ts_131072x1024x8192 gemv_f32_4096x65536 gemvtr_f32_65536x4096 reduce_f32_4096x65536 op_assign_f32_256M}
cd /tmp && export TMPDIR=/tmp
for wl in $WLS; do
  rm -rf $OUT/$wl
  steps=6; [ $wl = gemm_f16_32768 ] && steps=3
  WG_BENCH_NO_CHECK=1 rocprofv3 -i $ROOT/tools/pmc_r02.txt --kernel-trace --output-format csv -d $OUT/$wl -o p -- python3 $ROOT/bench.py --steps $steps --warmup 2 --workload $wl --no-secondary --no-cpu-baseline > $OUT/$wl.json 2> $OUT/$wl.log
done
cd $ROOT
python3 tools/pmc_r02_post.py $OUT $WLS
