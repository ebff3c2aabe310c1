#!/bin/bash
# Development aid (round 4): the exit-time crash of examples/strong_hip (SIGSEGV / glibc abort AFTER "Validated" and the timing line,
# about one run in ten on a GPU box).  Runs the driver under rocgdb until one run dies, at most $1 times (default 40), and prints the
# backtrace of every thread and the loaded libraries.  A host-side crash at process exit; the GPU work of every run completes.
n=${1:-40}; shift
args=${*:-64 32 8}
mkdir -p gpurun_out
for i in $(seq 1 $n); do
  timeout -k 5 60 /opt/rocm/bin/rocgdb -q -batch -ex "set pagination off" -ex "handle SIGUSR1 SIGUSR2 nostop noprint" -ex run -ex "bt 30" -ex "thread apply all bt 14" -ex "info sharedlibrary" \
    --args examples/strong_hip $args > gpurun_out/exit_crash_run.txt 2>&1
  if grep -q "SIGSEGV\|SIGABRT\|SIGBUS" gpurun_out/exit_crash_run.txt; then
    echo "== run $i died"; cp gpurun_out/exit_crash_run.txt gpurun_out/exit_crash_bt.txt
    grep -v "^\[New Thread\|^\[Thread .* exited\|^rank " gpurun_out/exit_crash_bt.txt | head -150
    exit 0
  fi
  echo "run $i: clean exit"
done
echo "no crash in $n runs under the debugger"
