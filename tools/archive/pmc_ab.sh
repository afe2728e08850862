#!/bin/bash
# cycles + clocks of library variants: tools/pmc_ab.sh <workload> <lib>...
wl=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  rm -rf gpurun_out/pmc_tmp
  WGEBRA_HIP_LIB=$PWD/wgmath_amd/$lib WG_BENCH_NO_CHECK=1 rocprofv3 -i tools/pmc_clk.txt --kernel-trace --output-format csv -d gpurun_out/pmc_tmp -o p -- python3 bench.py --steps 6 --warmup 2 --workload $wl --no-secondary --no-cpu-baseline > /dev/null 2>&1
  python3 - "$lib" <<'PY'
import csv, glob, sys, collections
agg=collections.defaultdict(list); dur=[]
for f in glob.glob('gpurun_out/pmc_tmp/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'gemm_f16_' in r['Kernel_Name'] or 'gemm_f32_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
m={k:sum(v)/len(v) for k,v in agg.items()}
d=sum(dur)/len(dur)
cyc=m.get('GRBM_GUI_ACTIVE',0)/8
print(f"{sys.argv[1]:28s} us={d:8.1f} cycles/XCD={cyc:.3e} clk={cyc/d/1e3:.3f}GHz mfma_busy/SIMD={m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024:.3e} util={m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/cyc:.3f} wave_cyc={m.get('SQ_WAVE_CYCLES',0)*4:.3e} wait_any={m.get('SQ_WAIT_ANY',0)*4:.3e} wait_inst={m.get('SQ_WAIT_INST_ANY',0)*4:.3e} active={m.get('SQ_ACTIVE_INST_ANY',0)*4:.3e}")
PY
done
