#!/bin/bash
# stall / issue breakdown of the f16 GEMM kernel: tools/pmc_stall.sh <workload>
wl=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_stall
WG_BENCH_NO_CHECK=1 rocprofv3 -i tools/pmc_stall.txt --kernel-trace --output-format csv -d gpurun_out/pmc_stall -o p -- python3 bench.py --steps 6 --warmup 2 --workload $wl --no-secondary --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_stall/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'gemm_f16_' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
wc=m.get('SQ_WAVE_CYCLES',1)
for k in sorted(m):
    print(f"{k:32s} {m[k]:14.4e}  {m[k]/wc*100:7.2f} % of SQ_WAVE_CYCLES")
PY
