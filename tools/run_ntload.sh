#!/bin/bash
# load-policy experiment on the 7-pt kernel: rebuild brk_star7.o with -DBRK_NT_OWN / -DBRK_NT_LAST on the box, kbench box mode
set -o pipefail
mkdir -p gpurun_out
{
for f in "" "-DBRK_NT_LAST" "-DBRK_NT_OWN" "-DBRK_NT_OWN -DBRK_NT_LAST"; do
  echo "== flags: $f"
  rm -f bricklib_amd/csrc/brk_star7.o
  make -C bricklib_amd/csrc EXTRA_HIPFLAGS="$f" 2>&1 | grep -E "error" 
  timeout -k 10 240 tools/kbench -n 512 -s 0 -S -b -v 0,0,0 -i 30; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi
done
} 2>&1 | tee gpurun_out/ntload.log
