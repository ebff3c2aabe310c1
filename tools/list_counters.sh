#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 120 rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
wc -l gpurun_out/counters_list.txt
grep -c "gfx950\|Name" gpurun_out/counters_list.txt
