#!/bin/bash
# 512^3 7-point sweep time and L2-miss traffic as a function of the traversal order (rows per block) and storage layout.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/sweep_order
mkdir -p $out
for lay in I S; do
  for o in 2 4 8; do
    a="-n 512 -s 0 -o $o -v 0 -i 20"; [ $lay = S ] && a="$a -S"
    timeout -k 10 120 tools/kbench $a > $out/t_${lay}_$o.log 2>&1 || { echo "kbench $lay $o failed"; tail -3 $out/t_${lay}_$o.log; exit 1; }
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_MISS_sum --output-format csv -d $out/p_${lay}_$o -- tools/kbench $a > $out/p_${lay}_$o.log 2>&1 || { echo "pmc $lay $o failed"; exit 1; }
    echo "== layout=$lay rows=$o"; grep -v "^#" $out/t_${lay}_$o.log | tail -2
    python3 - $out/p_${lay}_$o <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_s7" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("   ".join(f"{k}={sum(v[len(v)//2:])/len(v[len(v)//2:]):.4g}" for k, v in sorted(agg.items())))
PY
  done
done 2>&1 | tee $out/summary.txt
