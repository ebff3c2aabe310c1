#!/bin/bash
# round 2, run b: multi-rank rehearsal (gloo, shared GPU) + the weak workload on one GPU with the exchange forced through RCCL
set -o pipefail
mkdir -p gpurun_out/r02b
bash tools/rehearse_ranks.sh > gpurun_out/r02b/rehearse.txt 2>&1; rc=$?
tail -20 gpurun_out/r02b/rehearse.txt
if [ $rc -ne 0 ]; then echo "rehearsal failed"; exit 1; fi
for sch in plain surface deep; do
  for xm in packed direct shift; do
    timeout -k 10 300 python bench.py --workload weak --self-via-comm --schedule $sch --exchange $xm --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02b/weak_selfcomm_${sch}_${xm}.json 2> gpurun_out/r02b/weak_selfcomm_${sch}_${xm}.err; rc=$?
    if [ $rc -ne 0 ]; then echo "$sch $xm failed rc=$rc"; tail -5 gpurun_out/r02b/weak_selfcomm_${sch}_${xm}.err; exit 1; fi
    python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/r02b/weak_selfcomm_${sch}_${xm}.json') if l.startswith('{')][-1]); c=d['config']
print('$sch $xm', round(d['value'],1),'GSt/s ms/step',round(d['ms_per_step'],3),'exchange alone ms',round(c['exchange_ms_alone'],3),'msgs',c['messages_per_exchange'],'drift',c['conservation_rel_drift'],'transport',c['transport'])
assert c['conservation_rel_drift']<1e-12"
  done
done
timeout -k 10 300 python bench.py --workload weak --schedule plain --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02b/weak_links_plain.json 2>/dev/null
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r02b/weak_links_plain.json') if l.startswith('{')][-1]); c=d['config']
print('links plain', round(d['value'],1),'GSt/s ms/step',round(d['ms_per_step'],3),'exchange alone ms',round(c['exchange_ms_alone'],3))"
