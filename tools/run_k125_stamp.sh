#!/bin/bash
# in-kernel stamps of the cube kernels: rebuild libbrk.so with -DBRK_STAMP on the box (scratch copy), kbench variants
set -o pipefail
mkdir -p gpurun_out
{
rm -f bricklib_amd/csrc/brk_cubedense.o
make -C bricklib_amd/csrc EXTRA_HIPFLAGS=-DBRK_STAMP 2>&1 | tail -2
make -C tools 2>&1 | tail -1
for v in "$@"; do
  echo "== stencil/variants $v"
  timeout -k 10 240 tools/kbench -n 512 -s ${v%%:*} -S -o 8 -v ${v##*:} -i 10; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi
done
} 2>&1 | tee gpurun_out/k_stamp.log
