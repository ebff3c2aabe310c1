#!/bin/bash
# How much exchange time does each weak schedule hide?  One GPU, exchange stretched by a spin kernel on its stream.
set -o pipefail
mkdir -p gpurun_out/emu
for x in ${EMU_MS:-0 0.25 0.5 1 2}; do
  for s in plain surface deep auto; do
    timeout -k 10 200 python bench.py --workload weak --schedule $s --emulate-exchange-ms $x --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/emu/${s}_$x.json 2> gpurun_out/emu/${s}_$x.err || { echo "failed $s $x"; tail -3 gpurun_out/emu/${s}_$x.err; exit 1; }
  done
  python3 - $x <<'PY'
import json, sys
x = sys.argv[1]
row = []
for s in ("plain", "surface", "deep", "auto"):
    d = json.load(open(f"gpurun_out/emu/{s}_{x}.json"))
    row.append(f"{s} {d['ms_per_step']:.3f} ms" + (f" (picked {d['config']['schedule']}, probe {d['config']['schedule_probe_ms']:.2f})" if s == "auto" else ""))
print(f"exchange +{x} ms: " + " | ".join(row))
PY
done | tee gpurun_out/emu/summary.txt
