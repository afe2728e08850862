#!/bin/bash
# A/B timing of variant builds of the library: tools/ab.sh <workload> <lib1> <lib2> ...   (interleaved, 2 rounds)
wl=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    v=$(WGEBRA_HIP_LIB=$PWD/wgmath_amd/$lib WG_BENCH_NO_CHECK=1 python bench.py --steps 30 --warmup 5 --workload $wl --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
    echo "$lib $v"
  done
done
