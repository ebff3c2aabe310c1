#!/bin/bash
# round-3 kbench recipes: bash tools/run_r03.sh <recipe> ; output in gpurun_out/r03_<recipe>.txt
set -o pipefail
mkdir -p gpurun_out
K=tools/kbench
run() { echo "== kbench $*"; timeout -k 10 240 $K "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit 1; fi; }
case "$1" in
occ7)    # 7-point kernel (box and list form) at 24 / 16 / 8 waves per CU (padding with unused LDS)
  for L in 0 32000 100000; do run -n 512 -s 0 -S -b -v 0 -i 40 -L $L; run -n 512 -s 0 -S -o 8 -v 0,6 -i 40 -L $L; done ;;
occall)  # every stencil at reduced occupancy
  for s in 1 3 4 5; do for L in 0 8000 16000 24000; do run -n 512 -s $s -S -o 8 -v 0 -i 20 -L $L; done; done ;;
diag7)   # diagnostic build (make -C bricklib_amd/csrc EXTRA_HIPFLAGS=-DBRK_DIAG): what each re-read source of the 7-point box kernel costs
  export TMPDIR=/tmp
  for D in 0 1 2 3 4 8 13; do
    run -n 512 -s 0 -S -b -v 0 -i 60 -D $D
    rm -rf gpurun_out/diag7_pmc_$D
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/diag7_pmc_$D -- tools/kbench -n 512 -s 0 -S -b -v 0 -i 10 -D $D > /dev/null 2>&1
    python3 - $D <<'PY'
import csv, glob, sys, collections
D = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/diag7_pmc_{D}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_s7_wg" in r["Kernel_Name"] and "2" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    v = v[len(v) // 2:]
    print(f"   diag {D}: {k} mean over {len(v)} launches = {sum(v) / len(v):.6g}" + (f"  -> reads {2 * sum(v) / len(v) * 1024 / 1e9:.4f} GB" if k == "FETCH_SIZE" else ""))
PY
  done ;;
reg7d)   # diagnostic build: registered-adjacency mask path against its bounds (D 16 = mask path without the mask load, D 1 = no adjacency at all)
  run -n 512 -s 0 -S -b -v 0,100,300,200 -i 100 -r2 6 ;;
reg7)    # 7-point box kernel with / without brk_adjacency_register
  for r in 1 2 3; do run -n 512 -s 0 -S -b -v 0 -i 60; run -n 512 -s 0 -S -b -v 0 -i 60 -R; done ;;
order7)  # 7-point list form: rows per block of the XCD slab order
  for o in 2 4 8 16; do run -n 512 -s 0 -S -o $o -v 0 -i 60; done ;;
*) shift 0; run "$@" ;;
esac 2>&1 | tee gpurun_out/r03_$1.txt
