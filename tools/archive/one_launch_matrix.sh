#!/bin/bash
# same-box matrix: panel launches vs one launch per step x {tile scheduler on / off} x {write-through / plain stores (timing only)}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in libwgebra_hip.so libwgebra_hip_pplain.so; do
    for sched in -1 0; do
      echo "== rep $rep lib $lib sched $sched"
      WG_F16_SCHED=$sched WGEBRA_HIP_LIB=$PWD/wgmath_amd/$lib python tools/one_launch_ab.py 4 | tail -2
    done
  done
done
