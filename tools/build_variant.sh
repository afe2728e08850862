#!/bin/bash
# tools/build_variant.sh <name> <file.hip> "<-D flags>": libwgebra_hip_<name>.so = the current build with ONE translation unit recompiled
# with extra flags (A/B experiments; the variants travel to the GPU box with the snapshot and are loaded through WGEBRA_HIP_LIB).
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/../wgmath_amd/csrc"
make -j8 > /dev/null
mkdir -p _build_$name && cp _build/*.o _build_$name/
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -ffp-contract=on -w $flags -c $src -o _build_$name/${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwgebra_hip_$name.so _build_$name/*.o -ldl -lpthread
echo "built $name ($flags)"
