#!/bin/bash
# sample package power / sclk while a workload runs:  tools/power_probe.sh <workload> [lib]
wl=$1; lib=${2:-libwgebra_hip.so}
WGEBRA_HIP_LIB=$PWD/wgmath_amd/$lib WG_BENCH_NO_CHECK=1 python bench.py --steps ${STEPS:-3000} --warmup 5 --workload $wl --no-secondary --no-cpu-baseline > /tmp/pp.json 2>/dev/null &
pid=$!
sleep ${DELAY:-3}
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Package Power|sclk|junction|mclk" | sed 's/GPU\[0\]\s*: //' | tr '\n' '|'
  echo
  sleep 0.5
done
wait $pid
python3 -c "import json; d=json.load(open('/tmp/pp.json')); print('$wl $lib', d['value'], d['unit'], d['roofline']['kernel_ms'])"
