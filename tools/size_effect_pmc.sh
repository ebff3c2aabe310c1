#!/bin/bash
# round 5: what changes for the 13-point star between a 512^3 and a 1024^3 grid (3-4 % slower per brick)?  Memory-side and translation
# counters of the same kbench command at both sizes (k-paired lists, 8-row blocks): bash tools/size_effect_pmc.sh
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/size_pmc
for n in 512 1024; do
  it=$([ $n = 512 ] && echo 24 || echo 6)
  p=0
  for ctrs in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_sum" \
              "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "FETCH_SIZE" "WRITE_SIZE"; do     # (derived metrics: one per pass -- two together exceed the counter slots and rocprofv3 then waits to be killed)
    p=$((p + 1))
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/size_pmc/n${n}_p$p -- tools/kbench -n $n -s 3 -S -o 8 -PP -v 100 -i $it > gpurun_out/size_pmc/n${n}_p$p.log 2>&1
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "n=$n pass $p killed"; exit 1; fi
    if grep -q "exceeds the capabilities" gpurun_out/size_pmc/n${n}_p$p.log; then echo "n=$n pass $p: counter set refused"; exit 1; fi
    tail -1 gpurun_out/size_pmc/n${n}_p$p.log
  done
done
python3 - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/size_pmc/*/*/*counter_collection.csv")):
    n = re.search(r"/n(\d+)_p", f).group(1)
    for r in csv.DictReader(open(f)):
        if "k_star_km" in r["Kernel_Name"]:
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
ctrs = sorted({c for n in agg for c in agg[n]})
print("%-34s %16s %16s %10s   (per launch; 1024^3 has 8x the bricks)" % ("counter", "512^3", "1024^3", "ratio/8"))
for c in ctrs:
    a = sum(agg["512"][c]) / max(1, len(agg["512"][c])); b = sum(agg["1024"][c]) / max(1, len(agg["1024"][c]))
    print("%-34s %16.6g %16.6g %10.3f" % (c, a, b, b / a / 8 if a else float("nan")))
PY
