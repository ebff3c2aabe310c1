#!/bin/bash
# round 5: fabric read traffic (FETCH_SIZE, gfx950: x2 = bytes / 1024) of the star march kernels over k-paired lists, rows per block 8,
# with the rows of a block cut along i into chunks of C bricks (kbench -C; 0 = whole rows).  Counters in their own runs (kernel trace only).
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/ichunk_traffic
mkdir -p $out
for sid in 4 3; do
for c in 0 32 16; do
  a="-n 512 -s $sid -S -Q -o 8 -C $c -PP -v 100 -i 10 -r2 1"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p_${sid}_$c -- tools/kbench $a > $out/p_${sid}_$c.log 2>&1 || { echo "pmc $sid $c failed"; exit 1; }
  echo "== stencil $sid ichunk=$c"
  python3 - $out/p_${sid}_$c <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_star_km" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    m = sum(v[len(v)//2:]) / len(v[len(v)//2:])
    print(f"   {k} = {m:.5g} per launch -> fabric reads {2 * m * 1024 / 1e9:.3f} GB (x2 correction of gfx950)")
PY
done; done 2>&1 | tee $out/summary.txt
