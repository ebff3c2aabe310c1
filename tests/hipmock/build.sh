#!/bin/bash
# TEST INFRASTRUCTURE: builds, into a scratch directory, a CPU-executable and sanitized libbrk.so -- the HOST pass of every
# translation unit of bricklib_amd/csrc (clang++ -x hip --cuda-host-only, real HIP headers) + tests/hipmock/hipmock.cpp instead of
# libamdhip64 + the oracle's brick sweep as the emulated kernels -- a one-rank librccl.so.1 stand-in, the example drivers
# examples/strong_hip.cpp and weak_hip.cpp as they are, and tests/probe_asan_device_paths.cpp.
#   usage: tests/hipmock/build.sh OUTDIR        (needs /opt/rocm/lib/llvm/bin/clang++; no GPU)
set -euo pipefail
OUT=${1:?output directory}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CXX=${HIPMOCK_CXX:-/opt/rocm/lib/llvm/bin/clang++}
CC=${HIPMOCK_CC:-/opt/rocm/lib/llvm/bin/clang}
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -g -O1"
INC="-I$ROOT/include -I$ROOT/bricklib_amd/csrc -I/opt/rocm/include"
HIPHOST="-x hip --cuda-host-only --offload-arch=gfx950 -nogpulib -std=c++17 -fPIC -Wno-unused-command-line-argument -Wno-unknown-cuda-version"
mkdir -p "$OUT"
cd "$OUT"
# the two headers brk_jit.hip carries as text (bricklib_amd/csrc/Makefile: brk_jit_headers.inc)
{ echo 'const char kJitDeviceHeader[] = R"BRKSRC('; cat "$ROOT/include/brick-device.h"; echo ')BRKSRC";'
  echo 'const char kJitAbiHeader[] = R"BRKSRC('; cat "$ROOT/include/brk.h"; echo ')BRKSRC";'
  echo 'const char kJitSbaseHeader[] = R"BRKSRC('; cat "$ROOT/bricklib_amd/csrc/brk_sbase.h"; echo ')BRKSRC";'
  echo 'const char kJitStar7Header[] = R"BRKSRC('; cat "$ROOT/bricklib_amd/csrc/brk_star7_kernel.h"; echo ')BRKSRC";'
  echo 'const char kJitStarWgHeader[] = R"BRKSRC('; cat "$ROOT/bricklib_amd/csrc/brk_starwg_kernel.h"; echo ')BRKSRC";'
  echo 'const char kJitCubeHeader[] = R"BRKSRC('; cat "$ROOT/bricklib_amd/csrc/brk_cubedense_kernel.h"; echo ')BRKSRC";'; } > brk_jit_headers.inc
INC="-I$OUT $INC"
pids=()
for f in brk_kernels brk_star7 brk_starwg brk_cubedense brk_cube125q brk_place brk_exchange brk_jit; do
  $CXX $HIPHOST -cuid=$f $SAN $INC -c "$ROOT/bricklib_amd/csrc/$f.hip" -o $f.o &
  pids+=($!)
done
for f in brk_host brk_xplan; do
  $CXX -std=c++17 -fPIC $SAN $INC -c "$ROOT/bricklib_amd/csrc/$f.cpp" -o $f.o &
  pids+=($!)
done
$CXX $HIPHOST -cuid=hipmock $SAN $INC -c "$ROOT/tests/hipmock/hipmock.cpp" -o hipmock.o &
pids+=($!)
$CC -std=c11 -fPIC $SAN -c "$ROOT/oracle/brick_oracle.c" -o brick_oracle.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
# the host pass of a translation unit with kernels refers to its device code object as __hip_fatbin_<id>: there is none
nm brk_kernels.o brk_star7.o brk_starwg.o brk_cubedense.o brk_cube125q.o brk_place.o brk_exchange.o hipmock.o | awk '$1 == "U" && $2 ~ /^__hip_fatbin/ {print "const char " $2 "[16] = {0};"}' | sort -u > fatbins.c
$CC -fPIC -c fatbins.c -o fatbins.o
$CXX -shared $SAN -o libbrk.so brk_kernels.o brk_star7.o brk_starwg.o brk_cubedense.o brk_cube125q.o brk_place.o brk_exchange.o brk_jit.o brk_host.o brk_xplan.o hipmock.o fatbins.o brick_oracle.o -ldl -lm
$CXX -shared -fPIC -std=c++17 $SAN -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include "$ROOT/tests/hipmock/rcclmock.cpp" -o librccl.so.1
LINK="-L$OUT -lbrk -Wl,-rpath,$OUT"
$CXX $HIPHOST -cuid=strong_hip $SAN $INC -fopenmp -DBRK_NO_MPI -c "$ROOT/examples/strong_hip.cpp" -o strong_hip.o &
$CXX $HIPHOST -cuid=weak_hip $SAN $INC -DBRK_NO_MPI -c "$ROOT/examples/weak_hip.cpp" -o weak_hip.o &
$CXX $HIPHOST -cuid=probe $SAN $INC -DBRK_NO_MPI -c "$ROOT/tests/probe_asan_device_paths.cpp" -o probe.o &
wait
$CXX $SAN -fopenmp strong_hip.o -o strong_hip $LINK
$CXX $SAN weak_hip.o -o weak_hip $LINK
$CXX $SAN probe.o -o probe_asan_device_paths $LINK
echo "hipmock build: ok"
