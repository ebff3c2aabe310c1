#!/bin/bash
# PMC passes over kbench: bash prof_kbench.sh <tag> <kbench args...>
set -o pipefail
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/kpmc_$tag
run() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/kpmc_$tag/$name -- tools/kbench $KARGS > gpurun_out/kpmc_$tag/$name.log 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name killed"; exit 1; fi; }
KARGS="$*"
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run p2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVES
run p3 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr GRBM_GUI_ACTIVE
python3 - "$tag" <<'PY'
import csv, glob, sys, collections, re
tag=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"gpurun_out/kpmc_{tag}/*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(k_[a-z0-9_]+<[^>]*>)",r["Kernel_Name"])
        if m and not m.group(1).startswith("k_maxdiff") and not m.group(1).startswith("k_fill"):
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
names=sorted(agg); ctrs=sorted({c for k in agg for c in agg[k]})
with open(f"gpurun_out/kpmc_{tag}/summary.txt","w") as out:
    out.write("%-32s"%"counter"+"".join("%22s"%n[-20:] for n in names)+"\n")
    for c in ctrs:
        out.write("%-32s"%c+"".join("%22.5g"%(sum(agg[n].get(c,[0])[len(agg[n].get(c,[0]))//2:])/max(1,len(agg[n].get(c,[0])[len(agg[n].get(c,[0]))//2:]))) for n in names)+"\n")
print(open(f"gpurun_out/kpmc_{tag}/summary.txt").read())
PY
