#!/bin/bash
# bench.py over every stencil at 512^3 in both storage layouts -> gpurun_out/layouts.json
set -o pipefail
mkdir -p gpurun_out/layouts
for lay in separate interleaved; do
  for st in 7pt mpi7pt mpi13pt mpi25pt 27pt mpi125pt; do
    timeout -k 10 300 python bench.py --stencil $st --layout $lay --steps 100 --warmup 10 --no-cpu-baseline --no-weak-reference > gpurun_out/layouts/${lay}_$st.json 2> gpurun_out/layouts/${lay}_$st.err
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$st killed"; exit 1; fi
  done
done
python3 - <<'PY'
import json
out = {}
for lay in ("separate", "interleaved"):
    for st in ("7pt", "mpi7pt", "mpi13pt", "mpi25pt", "27pt", "mpi125pt"):
        d = json.load(open(f"gpurun_out/layouts/{lay}_{st}.json"))
        out.setdefault(st, {})[lay] = {"GStencil/s": round(d["value"], 1), "kernel_ms": round(d["roofline"]["kernel_ms"], 4),
                                       "frac_of_8TBs": round(d["roofline"]["frac"], 4), "max_abs_err_vs_oracle": d.get("max_abs_err_vs_oracle")}
json.dump(out, open("gpurun_out/layouts.json", "w"), indent=1)
for st, v in out.items():
    print(st, {k: (x["GStencil/s"], x["frac_of_8TBs"]) for k, x in v.items()})
PY
