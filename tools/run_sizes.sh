#!/bin/bash
# size sweep of every stencil through bench.py (each line carries max_abs_err_vs_oracle of a sampled oracle check): 64..768
set -o pipefail
mkdir -p gpurun_out/sizes
for n in 64 128 192 256 384 512 768; do
  for st in 7pt mpi13pt mpi25pt 27pt mpi125pt; do
    timeout -k 10 300 python bench.py --n $n --stencil $st --steps 20 --warmup 3 --windows 1 --steady-seconds 0 --no-config3 --no-cpu-baseline --no-weak-reference > gpurun_out/sizes/b_${st}_$n.json 2> gpurun_out/sizes/b_${st}_$n.err
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$st $n killed"; exit 1; fi
    python3 -c "import json;d=json.loads([l for l in open('gpurun_out/sizes/b_${st}_$n.json') if l.startswith('{')][-1]);print('$st', $n, round(d['value'],1),'GStencil/s', round(100*d['roofline']['frac'],1),'% err', d.get('max_abs_err_vs_oracle'))" || { tail -3 gpurun_out/sizes/b_${st}_$n.err; exit 1; }
  done
done 2>&1 | tee gpurun_out/sizes/summary.txt
