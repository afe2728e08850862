#!/bin/bash
# Regenerates the judged artefacts under profiles/ on the GPU box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench command          -> ${R}_bench_kernel_stats.csv, ${R}_kernel_durations_by_grid.csv,
#                                                                                ${R}_bench_under_rocprof.json
#   2. (counters: tools/pmc.sh -- one rocprofv3 run per workload, separate --pmc passes -> ${R}_pmc.csv, the source of roofline.traffic)
#   3. the plain bench line                                                   -> ${R}_bench.json
# Outputs go to gpurun_out/profiles_new/ (merged back by gpurun); copy them into profiles/ afterwards.
set -u
R=${R:-r06}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/profiles_new
rm -rf $OUT; mkdir -p $OUT
SKIP="--skip gemv_f32_1024_graph,gemv_f32_1024"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py --no-cpu-baseline --secondary-seconds 0.05 $SKIP > $OUT/${R}_bench_under_rocprof.json 2> $OUT/kt.log
cd $ROOT
python3 bench.py --detail $OUT/${R}_bench_detail.json > $OUT/${R}_bench.json 2> $OUT/bench.log
python3 tools/profiles_post.py $OUT $R
ls -la $OUT
