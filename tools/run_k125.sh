#!/bin/bash
# 125-point kernel variants side by side (kbench: each variant is checked against the generic kernel)
set -o pipefail
mkdir -p gpurun_out
K=tools/kbench
run() { echo "== kbench $*"; timeout -k 10 240 $K "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi; }
{
run -n 512 -s 5 -S -o 8 -v 1,0,10,11 -i 20
run -n 512 -s 5 -S -o 8 -v 10,0,10,0 -i 20
} 2>&1 | tee gpurun_out/k125.log
