#!/bin/bash
# Rehearsal of bench.py's multi-rank control flow on a ONE-GPU box: N processes share the GPU, ghost bytes go through host
# memory + gloo (bench.py --backend gloo).  Checks the conservation property, not speed.
set -o pipefail
mkdir -p gpurun_out/rehearse
port=29520
for cfg in "2 plain packed" "2 surface packed" "2 deep packed" "2 auto packed" "4 deep packed" "4 surface packed" "4 auto packed" "2 plain shift" "4 deep shift" "3 plain shift"; do   # at most 4 ranks: a one-GPU test box allows 6 processes on the card, launcher included
  set -- $cfg
  port=$((port + 1))
  out=gpurun_out/rehearse/n$1_$2_$3
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port \
    bench.py --gpus $1 --backend gloo --cells 256 --steps 3 --warmup 1 --no-cpu-baseline --schedule $2 --exchange $3 > $out.out 2> $out.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "N=$1 $2 FAILED rc=$rc"; tail -15 $out.err; exit 1; fi
  python3 - $out.out "$cfg" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
c = d["config"]
print(sys.argv[2], "cart", c["cart"], "schedule", c["schedule"], "remote MB", round(c["remote_bytes_per_rank"] / 1e6, 1),
      "msgs", c["messages_per_exchange"], "conservation drift", c["conservation_rel_drift"], "scaling_efficiency (rehearsal, not a measurement)", round(d["scaling_efficiency"], 3))
assert c["conservation_rel_drift"] < 1e-12
assert len(c["ranks_seen"]) == d["n_gpus"] and "scaling_base" in d
PY
done
# the star stencils of the weak workload: k-paired lists (13- and 25-point), i-chunked blocks (25-point)
for st in mpi13pt mpi25pt; do
  port=$((port + 1))
  out=gpurun_out/rehearse/n2_auto_$st
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
    bench.py --gpus 2 --backend gloo --cells 256 --steps 3 --warmup 1 --no-cpu-baseline --stencil $st > $out.out 2> $out.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "N=2 $st FAILED rc=$rc"; tail -15 $out.err; exit 1; fi
  python3 - $out.out "2 auto $st" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c = d["config"]
print(sys.argv[2], "schedule", c["schedule"], "conservation drift", c["conservation_rel_drift"], "kernel", d["roofline"].get("kernel", "")[:60])
assert c["conservation_rel_drift"] < 1e-12
PY
done
# strong workload: global 256^3 in 64^3 subdomains over 2 and 3 ranks
for np_ in 2 3; do
  port=$((port + 1))
  out=gpurun_out/rehearse/strong_n$np_
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $np_ --master-addr 127.0.0.1 --master-port $port \
    bench.py --gpus $np_ --backend gloo --workload strong --strong-dom 256 --strong-sub 64 --steps 3 --warmup 1 --no-cpu-baseline > $out.out 2> $out.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "strong N=$np_ FAILED rc=$rc"; tail -15 $out.err; exit 1; fi
  python3 - $out.out "strong $np_" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c = d["config"]
print(sys.argv[2], "subdomains/rank", c["subdomains_per_rank"], "remote MB", round(c["remote_bytes_per_rank"] / 1e6, 1), "conservation drift",
      c["conservation_rel_drift"])
assert c["conservation_rel_drift"] < 1e-12
PY
done
