#!/bin/bash
# round-2 GPU call: rank emulation (multi-GPU expectations), PMC summary, default bench line, kernel stats
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/rank_emulation.py 1 2 4 8 > $OUT/rank_emulation.json 2> $OUT/rank_emulation.err
tail -c 3000 $OUT/rank_emulation.json; tail -3 $OUT/rank_emulation.err
timeout 1500 bash tools/pmc_r02.sh > $OUT/pmc_r02.log 2>&1
tail -20 $OUT/pmc_r02.log
timeout 900 python bench.py > $OUT/r02_bench.json 2> $OUT/r02_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02_bench.json"))
print(d["metric"], d["value"], d["roofline"])
for o in d["others"]:
    print(o.get("workload"), o.get("value"), o.get("unit"), o.get("roofline",{}).get("frac"), o.get("roofline",{}).get("kernel_ms"), o.get("error"))
PY
