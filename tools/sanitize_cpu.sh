#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the code that runs on the CPU: the oracle (test infrastructure) rendering the
# seeded scenes of the test suite, and the HOST MIRROR of the product (from_box / from_obj / clip_and_project / Scene::project through
# the worker pool / the C API glue) projecting them.  No GPU is needed or used (GPU ASan is not available on the pool).
# usage: tools/sanitize_cpu.sh   -> prints "sanitizers: clean" and exits 0, or the first report
set -eu
cd "$(dirname "$0")/.."
OUT=build/sanitize
mkdir -p $OUT
SAN="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined"
g++ -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -pthread -Wall $SAN -shared -o $OUT/librusterix_oracle_asan.so oracle/rusterix_oracle.cpp oracle/oracle_capi.cpp
g++ -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -pthread -Wall $SAN -shared -Iinclude -o $OUT/librusterix_host_asan.so \
    rusterix_amd/csrc/host/rusterix_host.cpp rusterix_amd/csrc/host/host_capi.cpp -Lrusterix_amd/csrc -lrxr_hip -Wl,-rpath,$PWD/rusterix_amd/csrc
ASAN_LIB=$(g++ -print-file-name=libasan.so)
UBSAN_LIB=$(g++ -print-file-name=libubsan.so)
RXR_ORACLE_SO=$PWD/$OUT/librusterix_oracle_asan.so RXR_HOST_SO=$PWD/$OUT/librusterix_host_asan.so \
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$ASAN_LIB:$UBSAN_LIB" python3 tools/sanitize_driver.py "$@"
# ---- stage 2: the host code of librxr_hip.so at the shader boundary (hipcc: sanitizers on the HOST side only, -fno-gpu-sanitize)
if [ -x /opt/rocm/bin/hipcc ] && [ -f build/obj/rxr_kernels.hip.o ]; then
  CRT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
  HSAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-gpu-sanitize -shared-libsan"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC $HSAN -Iinclude -c -o $OUT/rxr_api_asan.o rusterix_amd/csrc/rxr_api.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC $HSAN -Iinclude -c -o $OUT/rxr_jit_asan.o rusterix_amd/csrc/rxr_jit.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $HSAN -o $OUT/librxr_hip_hostasan.so $OUT/rxr_api_asan.o $OUT/rxr_jit_asan.o \
      build/obj/rxr_multi.hip.o build/obj/rxr_kernels.hip.o build/obj/rxr_project.hip.o build/obj/rxr_selftest.hip.o -ldl
  RXR_DEVICE_SO=$PWD/$OUT/librxr_hip_hostasan.so ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  LD_PRELOAD="$CRT" python3 tools/sanitize_shaders.py
else
  echo "stage 2 skipped: no hipcc or no device objects (run __graft_entry__.build() first)"
fi

