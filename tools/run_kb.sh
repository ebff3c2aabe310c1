#!/bin/bash
# generic kbench runner: args are "sid:variants[:extra kbench flags]" (default flags: -S -o 8)
set -o pipefail
mkdir -p gpurun_out
make -C tools 2>&1 | tail -1
{
for s in "$@"; do
  IFS=: read sid vars extra <<< "$s"
  echo "== stencil $sid variants $vars ${extra:--S -o 8}"
  timeout -k 10 240 tools/kbench -n 512 -s $sid ${extra:--S -o 8} -v $vars -i 30; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi
done
} 2>&1 | tee gpurun_out/kb.log
