#!/bin/bash
# Usage (on the GPU box, from the repo root): bash run_gpu_checks.sh [tests|bench|prof|all]
# Every GPU step runs under its own timeout; a step that was killed stops the script.
set -o pipefail
mkdir -p gpurun_out
what=${1:-all}
killed() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }

if [ "$what" = all ] || [ "$what" = tests ]; then
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?
  tail -3 gpurun_out/smoke.log
  if killed $rc; then echo "smoke killed"; exit 1; fi
  timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout=300 > gpurun_out/pytest_gpu.log 2>&1; rc=$?
  tail -15 gpurun_out/pytest_gpu.log
  if killed $rc; then echo "pytest killed"; exit 1; fi
fi
if [ "$what" = all ] || [ "$what" = bench ]; then
  timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?
  cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
  if killed $rc; then echo "bench killed"; exit 1; fi
fi
if [ "$what" = all ] || [ "$what" = prof ]; then
  export TMPDIR=/tmp
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1; rc=$?
  tail -3 gpurun_out/prof_kt.log
  if killed $rc; then echo "prof killed"; exit 1; fi
  rm -rf gpurun_out/prof_kt.prev; ls -t $(find gpurun_out/prof_kt -name "*kernel_stats.csv") | head -1 | xargs cat | head -20   # newest run only
fi
