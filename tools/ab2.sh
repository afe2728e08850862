#!/bin/bash
# tools/ab2.sh <workload> <lib>...: every variant on zero operands (clock pinned: time = cycles) and on the bench's random operands
# (power-capped: what counts), interleaved, one round each. Prints TFLOP/s and kernel ms.
wl=$1; shift
for vals in zero rand; do
  for lib in "$@"; do
    v=$(WG_BENCH_VALUES=$vals WGEBRA_HIP_LIB=$PWD/wgmath_amd/$lib WG_BENCH_NO_CHECK=1 python bench.py --steps ${STEPS:-200} --warmup 20 --workload $wl --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
    echo "$vals $lib $v"
  done
done
