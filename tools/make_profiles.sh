#!/bin/bash
# Regenerates the judged artefacts under profiles/ on the GPU box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench command          -> r02_bench_kernel_stats.csv, r02_kernel_durations_by_grid.csv,
#                                                                                r02_bench_under_rocprof.json
#   2. rocprofv3 --pmc passes (tools/pmc_traffic.txt, separate run, no traces beyond --kernel-trace) -> traffic.json
#   3. the plain bench line                                                   -> r02_bench.json
# Outputs go to gpurun_out/profiles_new/ (merged back by gpurun); copy them into profiles/ afterwards.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/profiles_new
rm -rf $OUT; mkdir -p $OUT
SKIP="--skip gemv_f32_1024_graph,gemv_f32_1024"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py --no-cpu-baseline --secondary-seconds 0.05 $SKIP > $OUT/r02_bench_under_rocprof.json 2> $OUT/kt.log
rocprofv3 -i $ROOT/tools/pmc_traffic.txt --kernel-trace --output-format csv -d $OUT/pmc -o pmc -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --secondary-seconds 0.05 $SKIP > $OUT/pmc_bench.json 2> $OUT/pmc.log
cd $ROOT
python3 bench.py > $OUT/r02_bench.json 2> $OUT/bench.log
python3 tools/profiles_post.py $OUT
ls -la $OUT
