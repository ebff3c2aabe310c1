#!/bin/bash
# 25-pt timing experiment: what do the k-face / j-face loads cost?  Rebuilds libbrk.so with -DBRK_DIAG on the box (scratch copy):
# variants 7/8/9 drop the k- / j- / both face loads, 10/11/12 read them from the brick itself (L1 hits) -- wrong results on purpose.
set -o pipefail
mkdir -p gpurun_out
K=tools/kbench
run() { echo "== kbench $*"; timeout -k 10 240 $K "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi; }
{
rm -f bricklib_amd/csrc/brk_starwg.o
make -C bricklib_amd/csrc EXTRA_HIPFLAGS=-DBRK_DIAG 2>&1 | tail -1
run -n 512 -s 4 -S -o 8 -v 0,7,8,9 -i 20
run -n 512 -s 4 -S -o 8 -v 0,10,11,12 -i 20
} 2>&1 | tee gpurun_out/diag25.log
