#!/bin/bash
# Per-launch durations of one weak cycle under a given schedule (kernel trace only). Usage: bash tools/trace_weak.sh <schedule>
set -o pipefail
export TMPDIR=/tmp
s=${1:-deep}
out=gpurun_out/trace_$s
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --workload weak --schedule $s --no-cpu-baseline --steps 6 --warmup 2 > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
python3 tools/trace_cycle.py $out
