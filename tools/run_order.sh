#!/bin/bash
# traversal-order sweep: rows per block of the per-XCD slab order, per stencil (default kernels)
set -o pipefail
mkdir -p gpurun_out
K=tools/kbench
run() { echo "== kbench $*"; timeout -k 10 240 $K "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi; }
{
for s in 4 3 1 5; do
  for o in 8 4 2 16; do
    run -n 512 -s $s -S -o $o -v 0,0,0 -i 20
  done
done
} 2>&1 | tee gpurun_out/order.log
