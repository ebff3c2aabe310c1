#!/bin/bash
# PMC passes for the bench kernel (counters in their own runs, kernel-trace only; no sys/hip tracing).
# Usage: bash prof_pmc.sh <tag> [bench args...]
set -o pipefail
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/pmc_$tag
run() { # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag/$name -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-weak-reference --no-scripts --steady-seconds 0 --windows 1 $BENCH_ARGS > gpurun_out/pmc_$tag/$name.log 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name killed"; exit 1; fi
}
BENCH_ARGS="$*"
run p1 FETCH_SIZE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run p2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run p3 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
# round 4: the only memory-side counters this profiler lists that name DRAM -- L2 requests "destined for DRAM (MC)".  They sit on the
# L2 side of the fabric like RDREQ / WRREQ, i.e. IN FRONT of the Infinity Cache: compare a one-way run with `--alternate` to see
# whether they can tell Infinity-Cache hits from HBM reads (profiles/README.md, round 4)
run p4 TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
python3 - "$tag" <<'PY'
import csv, glob, sys, collections, re
tag=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{tag}/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(k_[a-z0-9_]+<[^>]*>)",r["Kernel_Name"])
        if m and ("k_star" in m.group(1) or "k_cube" in m.group(1) or "k_s7" in m.group(1)):
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
import json
traffic={}
with open(f"gpurun_out/pmc_{tag}/summary.txt","w") as out:
    for k,c in agg.items():
        out.write(k+"\n")
        for n,v in sorted(c.items()):
            out.write(f"  {n:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f,w=sum(c["FETCH_SIZE"])/len(c["FETCH_SIZE"]),sum(c["WRITE_SIZE"])/len(c["WRITE_SIZE"])
            # gfx950: FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; both in KB
            traffic[k]={"FETCH_SIZE_KB":f,"WRITE_SIZE_KB":w,"traffic_bytes_per_launch":(2*f+w)*1024}
            out.write(f"  -> traffic per launch (2 x FETCH_SIZE + WRITE_SIZE) = {(2*f+w)*1024/1e9:.4f} GB\n")
json.dump(traffic,open(f"gpurun_out/pmc_{tag}/traffic.json","w"),indent=1)
print(open(f"gpurun_out/pmc_{tag}/summary.txt").read())
PY
