#!/bin/bash
# GPU-side kernel durations of mid-size GEMMs (the Python probe is launch-bound below ~25 us): rocprofv3 --kernel-trace around
# tools/tall_skinny_probe.py; usage: tools/midsize_prof.sh <tag> <shape>...   (env WG_F16_TILE / WGEBRA_HIP_LIB / TR pass through)
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/midsize/$tag
rm -rf $OUT; mkdir -p $OUT
for shp in "$@"; do
  t=${shp//:/_}
  rocprofv3 --kernel-trace -d $OUT/$t -o p -- python3 $ROOT/tools/tall_skinny_probe.py $shp > $OUT/$t.log 2>&1
  python3 - $OUT/$t/p_results.db $shp <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
sym = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from {kd} d join {sym} s on d.kernel_id=s.id group by 1 having count(*) > 50 order by 3 desc"
print("==", sys.argv[2])
for r in con.execute(q):
    print("   %-58s n=%d avg=%.1f us min=%.1f" % (r[0][:58], r[1], r[2], r[3]))
PY
done
