#!/bin/bash
# sustained (200-step windows + 2 s steady state) timings through bench.py: args "stencil:variant"
set -o pipefail
mkdir -p gpurun_out/sustained
for sv in "$@"; do
  st=${sv%%:*}; v=${sv##*:}
  timeout -k 10 300 python bench.py --stencil $st --variant $v --no-cpu-baseline --no-config3 --no-weak-reference > gpurun_out/sustained/${st}_v$v.json 2>/dev/null; rc=$?
  if [ $rc -ne 0 ]; then echo "$sv failed"; exit 1; fi
  python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/sustained/${st}_v$v.json') if l.startswith('{')][-1])
print('$st variant $v: ms/step %.4f  steady kernel_ms %.4f  frac %.3f' % (d['ms_per_step'], d['steady_state']['kernel_ms'], d['steady_state']['roofline_frac']))"
done
