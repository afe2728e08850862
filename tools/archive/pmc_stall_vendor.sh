#!/bin/bash
# the same stall / issue counters (tools/pmc_stall.txt) for the vendor GEMM (torch.matmul -> hipBLASLt): tools/pmc_stall_vendor.sh <n> <layout>
n=$1; l=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_stall_v
rocprofv3 -i tools/pmc_stall.txt --kernel-trace --output-format csv -d gpurun_out/pmc_stall_v -o p -- python3 tools/vendor_gemm_one.py $n $l > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_stall_v/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'Cijk' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
wc=m.get('SQ_WAVE_CYCLES',1)
for k in sorted(m):
    print(f"{k:32s} {m[k]:14.4e}  {m[k]/wc*100:7.2f} % of SQ_WAVE_CYCLES")
PY
