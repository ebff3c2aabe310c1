#!/bin/bash
# box-mode kernels: parity tests for the box entry point, then timings; args "sid:variants" run box (-b) and list (-o 8) forms
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "box or full_size" --timeout=300 > gpurun_out/pytest_box.log 2>&1; rc=$?
tail -5 gpurun_out/pytest_box.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit 1; fi
{
for s in "$@"; do
  echo "== box  stencil/variants $s"
  timeout -k 10 240 tools/kbench -n 512 -s ${s%%:*} -S -b -v ${s##*:} -i 30; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi
  echo "== list stencil/variants $s"
  timeout -k 10 240 tools/kbench -n 512 -s ${s%%:*} -S -o 8 -v ${s##*:} -i 30; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi
done
} 2>&1 | tee gpurun_out/kbench_box.log
