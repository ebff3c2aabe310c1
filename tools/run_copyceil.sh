#!/bin/bash
# streaming ceilings on this box: brick-shaped copies (8-B and 16-B lanes, list / range / persistent) and the linear float4 copy
set -o pipefail
mkdir -p gpurun_out
{
timeout -k 10 240 tools/kbench -n 512 -s 0 -S -o 8 -c -v 0,0 -i 30
} 2>&1 | tee gpurun_out/copyceil.log
