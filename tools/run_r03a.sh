#!/bin/bash
# round 3, GPU job a: baseline (vendor vs ours), per-XCD tile timing at 8192^3 (stability over runs), epilogue time vs number of CUs storing at once
mkdir -p gpurun_out/r03a; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a
bash tools/vendor_vs_ours.sh > $O/vendor_vs_ours.txt 2>&1
T=$PWD/wgmath_amd/libwgebra_hip_trace1.so
for i in 1 2 3; do WGEBRA_HIP_LIB=$T python tools/f16_trace.py 8192 8192 8192 > $O/trace_8192_nn_$i.txt 2>&1; done
WGEBRA_HIP_LIB=$T python tools/f16_trace.py 8192 8192 8192 tn > $O/trace_8192_tn.txt 2>&1
WG_F16_TILE=256 WG_F16_NOSPLIT=1 WGEBRA_HIP_LIB=$T python tools/f16_trace.py 2048 8192 1024 > $O/trace_32tiles.txt 2>&1
WG_F16_TILE=256 WG_F16_NOSPLIT=1 WGEBRA_HIP_LIB=$T python tools/f16_trace.py 2048 8192 2048 > $O/trace_64tiles.txt 2>&1
WGEBRA_HIP_LIB=$T python tools/f16_trace.py 4096 8192 4096 > $O/trace_256tiles.txt 2>&1
WGEBRA_HIP_LIB=$T python tools/f16_trace.py 8192 8192 32768 > $O/trace_16rounds.txt 2>&1
