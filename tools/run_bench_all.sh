#!/bin/bash
# bench.py over every stencil of the path at 512^3 (config 2 and 3 of BASELINE.json) + kernel-trace stats
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/all
for st in 7pt mpi7pt mpi13pt mpi25pt 27pt mpi125pt; do
  timeout -k 10 300 python bench.py --stencil $st --steps 50 --warmup 5 --no-cpu-baseline --no-weak-reference > gpurun_out/all/bench_$st.json 2> gpurun_out/all/bench_$st.err
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$st killed"; exit 1; fi
  python3 -c "import json;d=json.load(open('gpurun_out/all/bench_$st.json'));print('$st', round(d['value'],1),'GStencil/s', round(d['roofline']['kernel_ms'],4),'ms', round(100*d['roofline']['frac'],1),'% of 8TB/s  err', d.get('max_abs_err_vs_oracle'))"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/all/prof27 -- python3 bench.py --stencil 27pt --steps 20 --warmup 3 --no-cpu-baseline --no-weak-reference > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/all/prof125 -- python3 bench.py --stencil mpi125pt --steps 20 --warmup 3 --no-cpu-baseline --no-weak-reference > /dev/null 2>&1
for p in prof27 prof125; do find gpurun_out/all/$p -name "*kernel_stats.csv" -exec head -3 {} \; | cut -c1-200; done
