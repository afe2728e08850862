#!/bin/bash
# round-2 first GPU call: copy-engine probe, launch floor, dispatch overhead (outputs under gpurun_out/)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
hipcc -O2 -std=c++17 -w -Iinclude tools/cpp/sdma_probe.cpp -o /tmp/sdma_probe -Lwgmath_amd -lwgebra_hip -lhsa-runtime64 -Wl,-rpath,$ROOT/wgmath_amd > $OUT/probe_build.log 2>&1
hipcc -O2 -std=c++17 -w --offload-arch=gfx950 tools/cpp/launch_floor.hip -o /tmp/launch_floor >> $OUT/probe_build.log 2>&1
g++ -O2 -std=c++17 -Iinclude tools/cpp/dispatch_overhead.cpp -o /tmp/dispatch_overhead wgmath_amd/libwgebra_hip.so -Wl,-rpath,$ROOT/wgmath_amd >> $OUT/probe_build.log 2>&1
timeout 300 /tmp/sdma_probe > $OUT/sdma_probe.txt 2>&1
timeout 120 /tmp/launch_floor > $OUT/launch_floor.txt 2>&1
timeout 120 /tmp/dispatch_overhead > $OUT/dispatch_overhead.txt 2>&1
cat $OUT/sdma_probe.txt $OUT/launch_floor.txt $OUT/dispatch_overhead.txt
