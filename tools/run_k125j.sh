#!/bin/bash
# 125-point: output-side j-shift variant (12) against the row-pipelined default (10) and the generic kernel (1)
set -o pipefail
mkdir -p gpurun_out
K=tools/kbench
run() { echo "== kbench $*"; timeout -k 10 240 $K "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi; }
{
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cube_kernel_variants and mpi125pt" --timeout=200 || exit 1
run -n 512 -s 5 -S -o 8 -v 1,10,12,12 -i 20
run -n 512 -s 5 -S -o 8 -v 12,10,12,10 -i 20
} 2>&1 | tee gpurun_out/k125j.log
