#!/bin/bash
# memory-side / dispatch-side PMC passes over kbench: bash prof_kbench2.sh <tag> <kbench args...>
set -o pipefail
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/kpmc2_$tag
run() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/kpmc2_$tag/$name -- tools/kbench $KARGS > gpurun_out/kpmc2_$tag/$name.log 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name killed"; exit 1; fi; }
KARGS="$*"
run q1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_LEVEL_WAVES SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE
run q2 SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_CSN_BUSY
run q3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run q5 TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
run q6 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TOTAL_CACHE_ACCESSES_sum
run q7 TCC_BUSY_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
run q8 TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
python3 - "$tag" <<'PY'
import csv, glob, sys, collections, re
tag=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"gpurun_out/kpmc2_{tag}/*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(k_[a-z0-9_]+<[^>]*>)",r["Kernel_Name"])
        if m and not m.group(1).startswith("k_maxdiff") and not m.group(1).startswith("k_fill"):
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
names=sorted(agg); ctrs=sorted({c for k in agg for c in agg[k]})
with open(f"gpurun_out/kpmc2_{tag}/summary.txt","w") as out:
    out.write("%-40s"%"counter"+"".join("%26s"%n[-24:] for n in names)+"\n")
    for c in ctrs:
        out.write("%-40s"%c+"".join("%26.6g"%(sum(agg[n].get(c,[0])[len(agg[n].get(c,[0]))//2:])/max(1,len(agg[n].get(c,[0])[len(agg[n].get(c,[0]))//2:]))) for n in names)+"\n")
print(open(f"gpurun_out/kpmc2_{tag}/summary.txt").read())
PY
