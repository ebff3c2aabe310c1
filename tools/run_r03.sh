#!/bin/bash
# round-3 kbench recipes: bash tools/run_r03.sh <recipe> ; output in gpurun_out/r03_<recipe>.txt
set -o pipefail
mkdir -p gpurun_out
K=tools/kbench
run() { echo "== kbench $*"; timeout -k 10 240 $K "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi; }
case "$1" in
occ7)    # 7-point kernel (box and list form) at 24 / 16 / 8 waves per CU (padding with unused LDS)
  for L in 0 32000 100000; do run -n 512 -s 0 -S -b -v 0 -i 40 -L $L; run -n 512 -s 0 -S -o 8 -v 0,6 -i 40 -L $L; done ;;
occall)  # every stencil at reduced occupancy
  for s in 1 3 4 5; do for L in 0 8000 16000 24000; do run -n 512 -s $s -S -o 8 -v 0 -i 20 -L $L; done; done ;;
*) shift 0; run "$@" ;;
esac 2>&1 | tee gpurun_out/r03_$1.txt
