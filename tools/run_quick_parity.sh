#!/bin/bash
# quick parity subset for kernel work: every stencil through list/range/box forms, padded lists, irregular lists
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stencil or box or padded or cube_kernel or star or fused or taps or golden" --timeout=300 > gpurun_out/pytest_quick.log 2>&1; rc=$?
tail -4 gpurun_out/pytest_quick.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit 1; fi
