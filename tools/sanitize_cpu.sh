#!/bin/bash
# CPU-side sanitizer runs (GPU AddressSanitizer is not available on this pool): the oracle under gcc ASAN + UBSAN, and the
# library's host code (host_io.cpp, is3d_run.cpp: parsers, writers, the run driver) under clang ASAN, with the CPU test files
# that exercise them.  Both leave the in-tree libraries as they found them.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
trap 'cp "$T/oracle_good.so" "$R/oracle/libcf_oracle.so" 2>/dev/null; cp "$T/lib_good.so" "$R/is3d_amd/lib/libis3d_amd.so" 2>/dev/null; rm -rf "$T"' EXIT
make -C "$R/oracle" >/dev/null
make -C "$R/is3d_amd/csrc" >/dev/null
cp "$R/oracle/libcf_oracle.so" "$T/oracle_good.so"
cp "$R/is3d_amd/lib/libis3d_amd.so" "$T/lib_good.so"
cd "$R"
echo "== oracle: gcc -fsanitize=address,undefined"
gcc -O1 -g -fopenmp -fPIC -std=gnu11 -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o "$R/oracle/libcf_oracle.so" oracle/cf_oracle.c -lm
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4 \
  LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  python -m pytest tests/test_oracle.py tests/test_oracle_sampler.py tests/test_oracle_vah.py tests/test_oracle_yield.py tests/test_oracle_dfcoef.py -x -q
cp "$T/oracle_good.so" "$R/oracle/libcf_oracle.so"
echo "== library host code: hipcc -x c++ -fsanitize=address (device objects unchanged)"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
C="$R/is3d_amd/csrc"; L="$R/is3d_amd/lib"
for f in host_io is3d_run; do $HIPCC -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-omit-frame-pointer -x c++ -c "$C/$f.cpp" -o "$T/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -pthread -fsanitize=address -shared-libasan -o "$L/libis3d_amd.so" "$L/cf_kernels.o" "$L/cf_feqmod.o" "$L/cf_sampler.o" "$L/cf_vah.o" "$L/cf_multi.o" "$L/cf_yield.o" "$L/cf_plan.o" "$T/host_io.o" "$T/is3d_run.o" -ldl
RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan*x86_64*.so" | head -1)
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:verify_asan_link_order=0 LD_PRELOAD=$RT python -m pytest tests/test_host_io.py tests/test_abi.py -x -q
