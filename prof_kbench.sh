#!/bin/bash
# PMC passes over kbench variants: bash prof_kbench.sh <tag> <kbench args...>
set -o pipefail
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/kpmc_$tag
run() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/kpmc_$tag/$name -- tools/kbench $KARGS > gpurun_out/kpmc_$tag/$name.log 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name killed"; exit 1; fi; }
KARGS="$*"
run p1 FETCH_SIZE TCC_HIT_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
run p2 WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
run p3 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag=sys.argv[1]
# dispatch order identifies the variant: group by (kernel name, dispatch index within name)
rows=[]
for f in sorted(glob.glob(f"gpurun_out/kpmc_{tag}/*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        rows.append(r)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r["Kernel_Name"]
    if "k_s7" in k or "k_star" in k:
        short=k.split("(")[0].replace("void brk::(anonymous namespace)::","").replace("void (anonymous namespace)::","")
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"gpurun_out/kpmc_{tag}/summary.txt","w") as out:
    for k,c in sorted(agg.items()):
        out.write(k+"\n")
        for n,v in sorted(c.items()):
            v2=v[len(v)//2:]  # skip warm-up half
            out.write(f"  {n:32s} n={len(v):3d} mean={sum(v2)/len(v2):.6g}\n")
print(open(f"gpurun_out/kpmc_{tag}/summary.txt").read())
PY
