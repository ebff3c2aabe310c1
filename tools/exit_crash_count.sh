#!/bin/bash
# Development aid (round 4): how often does examples/strong_hip die at exit, with and without the library the GPU pool preloads
# into every process?  Host-side, after all GPU work of a run has completed.
echo "LD_PRELOAD=[$LD_PRELOAD]"; echo "/etc/ld.so.preload: $(cat /etc/ld.so.preload 2>/dev/null)"
ls -la /usr/local/graft/lib/ 2>/dev/null | head
n=${1:-60}; shift
args=${*:-64 32 8}
for mode in preload nopreload; do
  bad=0; codes=""
  for i in $(seq 1 $n); do
    if [ $mode = nopreload ]; then env -u LD_PRELOAD examples/strong_hip $args > /dev/null 2>gpurun_out/exit_err.txt; rc=$?
    else examples/strong_hip $args > /dev/null 2>gpurun_out/exit_err.txt; rc=$?; fi
    if [ $rc -ne 0 ]; then bad=$((bad+1)); codes="$codes $rc"; tail -2 gpurun_out/exit_err.txt; fi
  done
  echo "$mode: $bad of $n runs of 'strong_hip $args' ended non-zero ($codes )"
done
