#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
K=tools/kbench
run() { echo "== kbench $*"; timeout -k 10 240 $K "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi; }
{
run -n 512 -s 5 -S -o 8 -v 1 -i 5
run -n 512 -s 1 -S -o 8 -v 1,0 -i 5
} 2>&1 | tee gpurun_out/kbench.log
