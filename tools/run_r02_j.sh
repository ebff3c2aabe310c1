#!/bin/bash
# round 2 evidence run (after the output-side 125-pt kernel): default bench line, kernel-trace stats of the same command, PMC passes (traffic per kernel)
set -o pipefail
mkdir -p gpurun_out/r02j
timeout -k 10 600 python bench.py > gpurun_out/r02j/bench_default.json 2> gpurun_out/r02j/bench_default.err; rc=$?
if [ $rc -ne 0 ]; then tail -5 gpurun_out/r02j/bench_default.err; exit 1; fi
python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r02j/bench_default.json') if l.startswith('{')][-1])
print('headline', round(d['value'],1), 'frac', round(d['roofline']['frac'],4), 'steady', round(d['steady_state']['value'],1))
for k,v in d['config3'].items():
    if isinstance(v,dict): print(' ',k, round(v['kernel_ms'],4),'ms frac',round(v['roofline']['frac'],3),'err',v.get('max_abs_err_vs_oracle'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
bash tools/run_gpu_checks.sh prof > gpurun_out/r02j/prof.log 2>&1; tail -12 gpurun_out/r02j/prof.log
cp $(ls -t $(find gpurun_out/prof_kt -name "*kernel_stats.csv") | head -1) gpurun_out/r02j/kernel_stats.csv
bash tools/prof_pmc.sh r02j > gpurun_out/r02j/pmc.log 2>&1; tail -40 gpurun_out/r02j/pmc.log
cp gpurun_out/pmc_r02j/summary.txt gpurun_out/r02j/pmc_summary.txt; cp gpurun_out/pmc_r02j/traffic.json gpurun_out/r02j/pmc_traffic_all.json
grep '^{' gpurun_out/prof_kt.log | tail -1 > gpurun_out/r02j/bench_under_rocprof.json
python3 - <<'PY'
import json, csv
d=json.loads(open('gpurun_out/r02j/bench_under_rocprof.json').read())
print('bench under rocprof: ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'))
for r in csv.DictReader(open('gpurun_out/r02j/kernel_stats.csv')):
    if 'k_s7_wg' in r['Name']: print('rocprof k_s7_wg calls', r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3)
PY
timeout -k 10 300 python bench.py --workload weak --no-cpu-baseline > gpurun_out/r02j/bench_weak_n1.json 2> gpurun_out/r02j/bench_weak_n1.err || tail -3 gpurun_out/r02j/bench_weak_n1.err
timeout -k 10 300 python bench.py --workload strong --no-cpu-baseline > gpurun_out/r02j/bench_strong_n1.json 2> gpurun_out/r02j/bench_strong_n1.err || tail -3 gpurun_out/r02j/bench_strong_n1.err
python3 - <<'PY'
import json
for f in ("weak","strong"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r02j/bench_{f}_n1.json") if l.startswith("{")][-1])
        print(f, round(d["value"],1), d["unit"], "ms_per_step", round(d["ms_per_step"],4), d["config"].get("workload"))
    except Exception as e: print(f, "failed", e)
PY
