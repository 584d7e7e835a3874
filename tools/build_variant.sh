#!/bin/bash
# Builds build/variants/librxr_hip_<name>.so from the working tree with extra compiler flags (A-B runs on the GPU box:
# tools/abab.sh, tools/try_variants.sh).   usage: tools/build_variant.sh name [-Dflags...]
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants build/obj_variant_$name
C=rusterix_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-unused-function -Iinclude"
pids=()
for f in rxr_api rxr_multi rxr_kernels rxr_project rxr_selftest rxr_jit; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c -o build/obj_variant_$name/$f.o $C/$f.hip 2>&1 | grep -i "error" -A5 &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o build/variants/librxr_hip_$name.so build/obj_variant_$name/*.o -ldl
rm -rf build/obj_variant_$name
ls -la build/variants/librxr_hip_$name.so
