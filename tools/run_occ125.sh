#!/bin/bash
# occupancy experiment (diagnostic build -DBRK_DIAG): the default 125-pt kernel at 12 / 8 / 4 waves per CU (unused dynamic LDS)
set -o pipefail
mkdir -p gpurun_out
{
rm -f bricklib_amd/csrc/brk_cubedense.o
make -C bricklib_amd/csrc EXTRA_HIPFLAGS=-DBRK_DIAG 2>&1 | tail -1
timeout -k 10 240 tools/kbench -n 512 -s 5 -S -o 8 -v 0,13,14 -i 20; rc=$?
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi
} 2>&1 | tee gpurun_out/occ125.log
