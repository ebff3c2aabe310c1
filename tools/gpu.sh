#!/bin/bash
# tools/gpu.sh -- ONE parametrised runner for everything that is run on the GPU box from the repository root
# (`gpurun -- 'bash tools/gpu.sh <recipe> [args]'`); output under gpurun_out/, what is worth keeping is copied to profiles/.
# Every GPU step runs under its own timeout and a step that was killed stops the script (no retries).
#   checks [tests|bench|prof|all]   smoke + pytest -m gpu / default bench line / rocprofv3 kernel-trace stats of the same command
#   census                          ONE run of examples/strong_hip 64 32 8 and of single_hip with BRK_CENSUS=1 (threads by id and name)
#   evidence <tag>                  end-of-round evidence: default bench line, kernel-trace stats and PMC passes of the same command
#                                   -> gpurun_out/<tag>/ (bench_default.json, kernel_stats.csv, pmc_summary.txt, pmc_traffic_all.json, ...)
#   strong                          strong tests + `bench.py --workload strong` with local ghosts aliased / copied + weak 25-pt cycle
#   expr                            stencils given as data: the GPU tests of the tap / expression paths, then tools/expr_bench.py (ms per sweep of
#                                   every form: tuned kernel, tap tables, interpreter, compiled at run time) -> gpurun_out/r05_expr.txt
#   kbench <kbench args>            tools/kbench with the arguments given
#   occ7 | occall | order7 | reg7   kernel experiments with the default library (occupancy padding, traversal order, registered adjacency)
#   diag7 | reg7d | abl | prio      the same with a diagnostic build (make -C bricklib_amd/csrc EXTRA_HIPFLAGS=-DBRK_DIAG before gpurun)
set -o pipefail
mkdir -p gpurun_out
K=tools/kbench
killed() { [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; }
run() { echo "== kbench $*"; timeout -k 10 240 $K "$@"; rc=$?; if killed $rc; then echo killed; exit 1; fi; }
recipe=${1:-checks}

checks() {
  what=${1:-all}
  if [ "$what" = all ] || [ "$what" = tests ]; then
    timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?
    tail -3 gpurun_out/smoke.log
    if killed $rc; then echo "smoke killed"; exit 1; fi
    timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_gpu.log 2>&1; rc=$?
    tail -15 gpurun_out/pytest_gpu.log
    if killed $rc; then echo "pytest killed"; exit 1; fi
  fi
  if [ "$what" = all ] || [ "$what" = bench ]; then
    timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?
    cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
    if killed $rc; then echo "bench killed"; exit 1; fi
  fi
  if [ "$what" = all ] || [ "$what" = prof ]; then
    export TMPDIR=/tmp
    rm -rf gpurun_out/prof_kt
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1; rc=$?
    tail -3 gpurun_out/prof_kt.log
    if killed $rc; then echo "prof killed"; exit 1; fi
    ls -t $(find gpurun_out/prof_kt -name "*kernel_stats.csv") | head -1 | xargs cat | head -8 | cut -c1-220
  fi
}

evidence() {
  tag=${1:?evidence needs a tag}; d=gpurun_out/$tag; mkdir -p $d
  timeout -k 10 600 python bench.py > $d/bench_default.json 2> $d/bench_default.err; rc=$?
  if [ $rc -ne 0 ]; then tail -5 $d/bench_default.err; exit 1; fi
  python3 - $d <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1] + '/bench_default.json') if l.startswith('{')][-1])
print('headline', round(d['value'], 1), 'frac', round(d['roofline']['frac'], 4), 'steady', round(d['steady_state']['value'], 1))
for k, v in d['config3'].items():
    if isinstance(v, dict): print(' ', k, round(v['kernel_ms'], 4), 'ms frac', round(v['roofline']['frac'], 3), 'err', v.get('max_abs_err_vs_oracle'))
print('cpu', d['cpu_baseline']['kind'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], '| port', (d.get('cpu_baseline_port') or {}).get('value'), '| alternating', (d.get('alternating') or {}).get('kernel_ms'))
PY
  checks prof > $d/prof.log 2>&1; tail -12 $d/prof.log
  cp $(ls -t $(find gpurun_out/prof_kt -name "*kernel_stats.csv") | head -1) $d/kernel_stats.csv
  bash tools/prof_pmc.sh $tag > $d/pmc.log 2>&1; tail -40 $d/pmc.log
  cp gpurun_out/pmc_$tag/summary.txt $d/pmc_summary.txt; cp gpurun_out/pmc_$tag/traffic.json $d/pmc_traffic_all.json
  # the same passes with successive sweeps in alternating directions: does any memory-side counter see the Infinity-Cache re-reads?
  bash tools/prof_pmc.sh ${tag}_alt --alternate --no-config3 > $d/pmc_alt.log 2>&1; tail -12 $d/pmc_alt.log
  cp gpurun_out/pmc_${tag}_alt/summary.txt $d/pmc_summary_alternating.txt; cp gpurun_out/pmc_${tag}_alt/traffic.json $d/pmc_traffic_alternating.json
  grep '^{' gpurun_out/prof_kt.log | tail -1 > $d/bench_under_rocprof.json
  python3 - $d <<'PY'
import json, csv, sys
d = json.loads(open(sys.argv[1] + '/bench_under_rocprof.json').read())
print('bench under rocprof: ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'))
for r in csv.DictReader(open(sys.argv[1] + '/kernel_stats.csv')):
    if 'k_s7_wg' in r['Name']: print('rocprof k_s7_wg calls', r['Calls'], 'avg us', float(r['AverageNs']) / 1e3, 'min', float(r['MinNs']) / 1e3)
PY
}

# ONE run of the strong driver with the thread census switched on (BRK_CENSUS=1: every thread of the process with its id and name at
# four points of main(), examples/crash_trace.h) -- who owns tid = pid + 1, the thread that died in round 4
census() {
  for exe in "strong_hip 64 32 8" "single_hip"; do
    echo "== BRK_CENSUS=1 examples/$exe"
    BRK_CENSUS=${BRK_CENSUS_LEVEL:-1} timeout -k 10 240 examples/$exe 2>&1 | grep -v '^rank [1-7]:' ; rc=${PIPESTATUS[0]}
    echo "== exit status $rc"
    if killed $rc; then echo killed; exit 1; fi
  done
  echo "== LD_PRELOAD=$LD_PRELOAD"
}

case "$recipe" in
checks) shift; checks "$@" ;;
census) census 2>&1 | tee gpurun_out/r05_census.txt ;;
evidence) shift; evidence "$@" ;;
kbench) shift; run "$@" 2>&1 | tee gpurun_out/kbench.log ;;
*) { case "$recipe" in

strong) # strong workload (config 5 on one GPU): ghosts between subdomains of one rank aliased through the adjacency table vs copied
  timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_cpp_headers.py -m gpu -q -x --timeout=600 -k "strong or kpaired" > gpurun_out/pytest_strong.log 2>&1; rc=$?
  tail -5 gpurun_out/pytest_strong.log
  if [ $rc -ne 0 ]; then exit 1; fi
  for g in aliased copied; do
    timeout -k 10 400 python bench.py --workload strong --strong-local-ghosts $g --no-cpu-baseline > gpurun_out/strong_$g.json 2> gpurun_out/strong_$g.err; rc=$?
    if [ $rc -ne 0 ]; then tail -5 gpurun_out/strong_$g.err; exit 1; fi
    python3 - gpurun_out/strong_$g.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(d['config'].get('local_ghosts'), round(d['value'], 1), d['unit'], 'ms/step', round(d['ms_per_step'], 3), 'swept', d['config'].get('swept_bricks_per_sweep'),
      'of', d['config'].get('subdomain_bricks'), 'drift', d['config'].get('conservation_rel_drift'))
PY
  done
  for g in alias copy; do timeout -k 10 200 examples/strong_hip 512 128 1 $g | tail -2; done
  timeout -k 10 300 python bench.py --workload weak --stencil mpi25pt --no-cpu-baseline > gpurun_out/weak_25pt.json 2> gpurun_out/weak_25pt.err; rc=$?
  tail -1 gpurun_out/weak_25pt.json | cut -c1-400
  ;;

l1slots) # outstanding L1->L2 read requests per TCP in the brick-copy probe (tools/l1probe): latency sum / (kernel cycles x 256)
  export TMPDIR=/tmp
  for l in 0 53000 80000; do
    rm -rf gpurun_out/l1slots_$l
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/l1slots_$l -- tools/l1probe -w 4 -l $l -i 6 > /dev/null 2>&1
    python3 - $l <<'PY'
import csv, glob, sys, collections, re
l = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/l1slots_{l}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_probe<([^>]*)>", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    g = lambda n: sum(c[n]) / max(1, len(c[n]))
    cyc = g("GRBM_GUI_ACTIVE") / 8
    if g("TCP_TCC_READ_REQ_sum"):
        print(f"extra LDS {l}: k_probe<{k}> (own, halo, planes/side, waves): {g('TCP_TCC_READ_REQ_sum')/256:.0f} requests per TCP, latency {g('TCP_TCC_READ_REQ_LATENCY_sum')/g('TCP_TCC_READ_REQ_sum'):.0f} clk, "
              f"kernel {cyc:.0f} clk -> {g('TCP_TCC_READ_REQ_LATENCY_sum')/256/cyc:.1f} requests in flight per TCP")
PY
  done ;;
tapmc)   # how busy is the vector memory path (TA / TCP / TD) in each default kernel?  PMC passes over kbench, one stencil per process
  export TMPDIR=/tmp
  for sid in 0 3 4 1 5; do
    i=0
    for ctrs in "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
                "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
                "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
      i=$((i + 1)); rm -rf gpurun_out/tapmc_${sid}_$i
      timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/tapmc_${sid}_$i -- tools/kbench -n 512 -s $sid -S -Q -b -v 100 -i 10 > /dev/null 2>&1
    done
    python3 - $sid <<'PY'
import csv, glob, sys, collections
sid = sys.argv[1]
agg = collections.defaultdict(list)
name = None
for f in glob.glob(f"gpurun_out/tapmc_{sid}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if (("k_s7_wg<0" in k and sid == "0") or "k_star_wg" in k or "k_star_km" in k or "k_cube_dense" in k):
            name = k.split("(")[0][-40:]
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"stencil {sid}: {name}")
for k, v in sorted(agg.items()):
    v = v[len(v) // 2:]
    print(f"   {k:44s} {sum(v) / len(v):.6g}")
PY
  done ;;
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
abl)     # diagnostic build: ablations of the cube kernels (variant code = 1000 + 100 * bits): what does each part of a wave's work cost?
  for sid in 1 5; do run -n 512 -s $sid -S -o 8 -v 0,1100,1200,1400,1800,2600,4200,1500,1700,4700,7300 -i 30 -r2 3; done ;;
prio)    # diagnostic build: compute phase (bit 64) / fetch phase (bit 128) of the cube kernels at raised priority
  for sid in 5 1 7; do run -n 512 -s $sid -S -o 8 -v 0,7400,13800 -i 40 -r2 4; done ;;
reg7d)   # diagnostic build: registered-adjacency mask path against its bounds (D 16 = mask path without the mask load, D 1 = no adjacency at all)
  run -n 512 -s 0 -S -b -v 0,100,300,200 -i 100 -r2 6 ;;
kpair)   # star kernels: two bricks per wave over k-paired lists (variant 3) against the default kernel on its tiled list
  for sid in 4 3; do
    run -n 512 -s $sid -S -o 8 -v 0 -i 40 -r2 3
    for o in 8 4 2; do run -n 512 -s $sid -S -o $o -P -v 3,0 -i 40 -r2 3; done
  done ;;
regstar) # star kernels (list form) with / without brk_adjacency_register: compact face table instead of the 108-byte rows
  for sid in 0 3 4; do run -n 512 -s $sid -S -o 8 -v 0,100 -i 40 -r2 4; done ;;
regcube) # cube kernels (list form) with / without brk_adjacency_register: rows of the list neighbours replaced by link bits
  for sid in 1 5 7; do run -n 512 -s $sid -S -o 8 -v 0,100 -i 40 -r2 4; done ;;
orderx2) # fused double 7-point pass (list form): rows per block of the XCD slab order
  for o in 2 4 8; do run -n 512 -s 7 -S -o $o -v 100 -i 40 -r2 2; done ;;
zigzag)  # every other sweep walks the bricks backwards (what the last sweep touched last is read first: Infinity-Cache reuse)
  for sid in 0 3 7; do run -n 512 -s $sid -S -o 8 -v 100 -i 60 -r2 2; run -n 512 -s $sid -S -o 8 -v 100 -i 60 -r2 2 -Z; done
  for w in "" "--one-way"; do
    timeout -k 10 300 python bench.py --no-cpu-baseline $w > gpurun_out/zz_single$w.json 2> gpurun_out/zz_single$w.err || { tail -5 gpurun_out/zz_single$w.err; exit 1; }
    timeout -k 10 300 python bench.py --workload weak --no-cpu-baseline --no-scaling-base $w > gpurun_out/zz_weak$w.json 2> gpurun_out/zz_weak$w.err || { tail -5 gpurun_out/zz_weak$w.err; exit 1; }
    timeout -k 10 400 python bench.py --workload strong --no-cpu-baseline $w > gpurun_out/zz_strong$w.json 2> gpurun_out/zz_strong$w.err || { tail -5 gpurun_out/zz_strong$w.err; exit 1; }
    python3 - "$w" <<'PY'
import json, sys
w = sys.argv[1]
for k in ("single", "weak", "strong"):
    d = json.loads([l for l in open(f"gpurun_out/zz_{k}{w}.json") if l.startswith("{")][-1])
    extra = ""
    if k == "single":
        extra = " one_way %s | " % (d.get("one_way") or {}).get("kernel_ms") + " ".join("%s %.4f/%.4f" % (n, v["kernel_ms"], v.get("kernel_ms_one_way", 0)) for n, v in d.get("config3", {}).items() if isinstance(v, dict))
    print(w or "alternating", k, round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), extra)
PY
  done ;;
km13)    # 13-point march at 5 waves per SIMD (variant 5, dev build) against the default march (variant 0 over a k-paired list: kbench declares nothing, so the tiled kernel answers -- use bench for the default)
  run -n 512 -s 3 -S -o 8 -P -v 105,103,100 -i 60 -r2 3
  run -n 512 -s 4 -S -o 8 -P -v 105,103 -i 40 -r2 2 ;;
km7)     # 7-point: two k-consecutive bricks per wave with ONE tile (variant 10: 8-wave groups, 11: 4-wave groups; dev build) vs the default
  run -n 512 -s 0 -S -o 8 -v 100 -i 60 -r2 3
  run -n 512 -s 0 -S -o 8 -P -G 8 -v 110,112,100 -i 60 -r2 3
  run -n 512 -s 0 -S -o 8 -P -G 4 -v 111,100 -i 60 -r2 3
  run -n 512 -s 2 -S -o 8 -P -G 8 -v 110,100 -i 40 -r2 2
  run -n 512 -s 0 -S -o 8 -v 110 -i 20 -r2 1 ;;
kmarch)  # star kernels: NB k-consecutive bricks per wave (variant 4: NB = 4, 5: NB = 2; dev build) over k-stacked lists vs the defaults
  for sid in 4 3; do
    run -n 512 -s $sid -S -o 8 -v 100 -i 40 -r2 2
    run -n 512 -s $sid -S -o 8 -P -v 103,105 -i 40 -r2 3
    run -n 512 -s $sid -S -o 8 -P -K 4 -v 104 -i 40 -r2 3
    run -n 512 -s $sid -S -o 4 -P -K 4 -v 104 -i 40 -r2 2
    run -n 512 -s $sid -S -o 8 -v 104 -i 20 -r2 1     # the slow path (lists that are not k-stacked): correctness + cost
  done ;;
pair7)   # 7-point: two bricks per wave (variants 7 / 8 / 9, dev build) over k-paired and j-paired lists against the default kernel (list form, registered)
  run -n 512 -s 0 -S -o 8 -v 100 -i 60 -r2 3
  for ax in -P -J; do
    run -n 512 -s 0 -S -o 8 $ax -G 4 -v 107,108,100 -i 60 -r2 3
    run -n 512 -s 0 -S -o 8 $ax -G 8 -v 109,100 -i 60 -r2 3
  done
  run -n 512 -s 0 -S -o 4 -J -G 4 -v 107 -i 60 -r2 2
  run -n 512 -s 0 -S -o 2 -J -G 4 -v 107 -i 60 -r2 2
  run -n 512 -s 0 -S -b -v 100 -i 60 -r2 3 ;;
predrow) # DEV=1 build: cube kernels, box entry point over a registered table -- adjacency rows computed for bricks whose 26 neighbours are the predicted ids (100) vs every row loaded (125600)
  for sid in 1 5 7; do run -n 512 -s $sid -S -b -v 100,125600 -i 40 -r2 4; done ;;
shift7)  # 7-point box kernel: the output storage shifted against the input storage (does the DRAM channel / bank alignment of the two streams matter?)
  for A in ${SHIFTS:-0 256 1024 4096 8192 32768 69632 1052672 0}; do run -n 512 -s ${SID:-0} -S -b -v 100 -i ${IT:-100} -r2 2 ${AOPT:--A} $A; done ;;
place7)  # 7-point box kernel over freshly allocated output storages in ONE process: does the physical placement matter?
  run -n 512 -s ${SID:-0} -S -b -v 100 -i ${IT:-60} ${TOPT:--T} ${TRIALS:-24} ${FOPT:-} ;;
tune7p)  # DEV=1 build, storages placed (-Q): the 7-point kernel's knobs once more, now that the pair of storages is of the fast kind
  run -n 512 -s 0 -S -Q -b -v 100 -i 100 -r2 3
  for L in 0 32000; do run -n 512 -s 0 -S -Q -b -v 100 -i 100 -r2 2 -L $L; done
  for o in 2 4 8 16; do run -n 512 -s 0 -S -Q -o $o -v 100 -i 100 -r2 2; done
  run -n 512 -s 0 -S -Q -b -v 100 -i 100 -r2 2 -Z ;;
rows25)  # star kernels (march, k-paired lists declared) over placed storages: rows per block of the XCD slab order, time and (below) traffic
  for sid in 4 3; do for o in 8 4 2; do run -n 512 -s $sid -S -Q -o $o -PP -v 100 -i 60 -r2 2; done; done ;;
dmacube) # VARIANTS=1 build: cube kernels staged by LDS-DMA (variant 15, k_cube_dma) vs through registers (0), registered table, placed storages, box and list form
  for sid in 1 5 7; do run -n 512 -s $sid -S -Q -b -v 115,100 -i 40 -r2 4; run -n 512 -s $sid -S -Q -o 8 -v 115,100 -i 40 -r2 3; done ;;
q125)    # VARIANTS=1 build, round 5: the lean 125-point kernel at 4 waves per SIMD (variant 16: timing diagnostic, wrong i-ends) vs the default (0), registered table, placed storages
  run -n 512 -s 5 -S -Q -b -v 116,117,100 -i 40 -r2 4; run -n 512 -s 5 -S -Q -o 8 -v 116,117,100 -i 40 -r2 3 ;;
dpp7)    # VARIANTS=1 build, round 5: 7-point kernel with the +-i neighbours by DPP (variant 13) vs the default (0), list form, placed storages
  for sid in 0 2; do run -n 512 -s $sid -S -Q -o 8 -v 113,100 -i 100 -r2 4; done ;;
ichunk)  # round 5: rows of a block cut along i into chunks (kbench -C): the k-reuse distance of the k-paired star lists halves without fewer rows per block.  Default build.
  for sid in 4 3; do for c in 0 32 16; do run -n 512 -s $sid -S -Q -o 8 -C $c -PP -v 100 -i 60 -r2 2; done; done
  for sid in 1 5 7 0; do for c in 0 32; do run -n 512 -s $sid -S -Q -o 8 -C $c -v 100 -i 40 -r2 2; done; done ;;
expr)    # round 5: the expression path (non-linear scripts): new GPU tests, then ms per 512^3 sweep beside the tuned cond kernel and the tap path
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_cpp_headers.py -m gpu -q -x --timeout=300 -k "expression or single_hip or compiled or taps" > gpurun_out/pytest_expr.log 2>&1; rc=$?
  tail -5 gpurun_out/pytest_expr.log
  if [ $rc -ne 0 ]; then exit 1; fi
  timeout -k 10 300 python tools/expr_bench.py ;;
big13)   # round 5: the 13-point star on ONE 1024^3 grid (rows of 128 bricks: a block of 8 rows is 4 MiB = the L2 of an XCD): i-chunks of 64 / 32 bricks vs none; 512^3 beside it
  for c in 0 64 32; do run -n 1024 -s 3 -S -o 8 -C $c -PP -v 100 -i 20 -r2 2; done
  for c in 0 32; do run -n 512 -s 3 -S -o 8 -C $c -PP -v 100 -i 60 -r2 2; done ;;
reg7)    # 7-point box kernel with / without brk_adjacency_register
  for r in 1 2 3; do run -n 512 -s 0 -S -b -v 0 -i 60; run -n 512 -s 0 -S -b -v 0 -i 60 -R; done ;;
order7)  # 7-point list form: rows per block of the XCD slab order
  for o in 2 4 8 16; do run -n 512 -s 0 -S -o $o -v 0 -i 60; done ;;

*) echo "unknown recipe $recipe"; exit 2 ;;
esac; } 2>&1 | tee gpurun_out/${RTAG:-r05}_$recipe.txt ;;
esac
