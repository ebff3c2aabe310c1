#!/bin/bash
# quick look at the cube kernels after a change: parity of every cube variant, then kbench 27-pt / 125-pt defaults
set -o pipefail
mkdir -p gpurun_out
K=tools/kbench
run() { echo "== kbench $*"; timeout -k 10 240 $K "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi; }
{
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cube_kernel_variants or fused or irregular or box" --timeout=200 || exit 1
run -n 512 -s 1 -S -o 8 -v 0,0,0,0 -i 20
run -n 512 -s 5 -S -o 8 -v 0,10,0,10 -i 20
run -n 512 -s 7 -S -o 8 -v 0,0,0 -i 20
} 2>&1 | tee gpurun_out/cubeq.log
