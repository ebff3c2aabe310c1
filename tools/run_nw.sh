#!/bin/bash
# group size re-check for the star kernels: 25-pt 2 (default) vs 4 waves (variant 2), 13-pt 4 (default) vs 8 waves (variant 2)
set -o pipefail
mkdir -p gpurun_out
K=tools/kbench
run() { echo "== kbench $*"; timeout -k 10 240 $K "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi; }
{
run -n 512 -s 4 -S -o 8 -v 0,2,0,2 -i 20
run -n 512 -s 3 -S -o 8 -v 0,2,0,2 -i 20
} 2>&1 | tee gpurun_out/nw.log
