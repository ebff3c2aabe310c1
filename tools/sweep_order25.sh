#!/bin/bash
# 25-point sweep: fabric read traffic (FETCH_SIZE) vs rows per block
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/sweep_order25
mkdir -p $out
for o in 8 4 2; do
  a="-n 512 -s 4 -S -o $o -v 0 -i 10"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_MISS_sum --output-format csv -d $out/p_$o -- tools/kbench $a > $out/p_$o.log 2>&1 || { echo "pmc $o failed"; exit 1; }
  echo "== rows=$o"
  python3 - $out/p_$o <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_star_wg" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("   ".join(f"{k}={sum(v[len(v)//2:])/len(v[len(v)//2:]):.4g}" for k, v in sorted(agg.items())))
PY
done 2>&1 | tee $out/summary.txt
