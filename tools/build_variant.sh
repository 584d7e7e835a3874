#!/bin/bash
# Builds build/variants/librxr_hip_<name>.so from the working tree with extra compiler flags.   usage: tools/build_variant.sh name [-Dflags...]
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
C=rusterix_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wno-unused-function -Iinclude "$@" \
  -o build/variants/librxr_hip_$name.so $C/rxr_api.hip $C/rxr_kernels.hip $C/rxr_project.hip $C/rxr_selftest.hip 2>&1 | grep -i "error" -A5
ls -la build/variants/librxr_hip_$name.so
