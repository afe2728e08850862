#!/bin/bash
# kernel-level breakdown of mid-size GEMMs (which kernels, how long): rocprofv3 --kernel-trace --stats around tall_skinny_probe.py
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/midsize
rm -rf $OUT; mkdir -p $OUT
for shp in f16:2048x2048x2048 f16:1024x1024x1024 f16:3072x3072x3072 f32:2048x2048x2048 f32:3072x3072x3072 f32:1024x1024x1024; do
  tag=${shp//:/_}
  rocprofv3 --kernel-trace --stats -d $OUT/$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/tall_skinny_probe.py $shp > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name '*kernel_stats.csv' | head -1)
  echo "== $shp"; tail -1 $OUT/$tag.log; head -8 "$f" | cut -d, -f1-8
done
