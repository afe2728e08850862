#!/bin/bash
# PMC view of the vendor GEMM (torch.matmul -> hipBLASLt) next to ours: tools/pmc_vendor.sh <n> <layout nn|nt|tn>
n=$1; l=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pmc in pmc_clk pmc_traffic; do
  rm -rf gpurun_out/pmc_v
  rocprofv3 -i tools/$pmc.txt --kernel-trace --output-format csv -d gpurun_out/pmc_v -o p -- python3 tools/vendor_gemm_one.py $n $l > /dev/null 2>&1
  python3 - "$l" <<'PY'
import csv, glob, sys, collections
agg=collections.defaultdict(list); dur=[]
for f in glob.glob('gpurun_out/pmc_v/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'Cijk' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
m={k:sum(v)/len(v) for k,v in agg.items()}
d=sum(dur)/len(dur)
if 'GRBM_GUI_ACTIVE' in m:
    cyc=m['GRBM_GUI_ACTIVE']/8
    print(f"vendor {sys.argv[1]} us={d:8.1f} cycles/XCD={cyc:.3e} clk={cyc/d/1e3:.3f}GHz util={m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/cyc:.3f} wave_cyc={m.get('SQ_WAVE_CYCLES',0)*4:.3e} wait_any={m.get('SQ_WAIT_ANY',0)*4:.3e} wait_inst={m.get('SQ_WAIT_INST_ANY',0)*4:.3e} active={m.get('SQ_ACTIVE_INST_ANY',0)*4:.3e} lds_inst={m.get('SQ_ACTIVE_INST_LDS',0)*4:.3e}")
else:
    print(f"vendor {sys.argv[1]} us={d:8.1f} " + " ".join(f"{k}={v:.4g}" for k,v in m.items()))
PY
done
