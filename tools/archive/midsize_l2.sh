#!/bin/bash
# L2 hit rate of the mid-size GEMM kernels: tools/midsize_l2.sh <shape>...
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/l2tmp
for shp in "$@"; do
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 -i $ROOT/tools/pmc_l2.txt --kernel-trace --output-format csv -d $OUT -o p -- python3 $ROOT/tools/tall_skinny_probe.py $shp > $OUT/log 2>&1
  python3 - $OUT $shp <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'gemm_f16' in k or 'Cijk' in k:
            agg[k[:50]][r['Counter_Name']].append(float(r['Counter_Value']))
print("==", sys.argv[2])
for k, m in agg.items():
    a = {c: sum(v) / len(v) for c, v in m.items()}
    hit, miss = a.get('TCC_HIT_sum', 0), a.get('TCC_MISS_sum', 0)
    print("   %-50s req=%.3e hit=%.3e miss=%.3e hit_rate=%.3f" % (k, a.get('TCC_REQ_sum', 0), hit, miss, hit / max(hit + miss, 1)))
PY
done
rm -rf $OUT
