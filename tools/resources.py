#!/usr/bin/env python3
"""Condense hipcc -Rpass-analysis=kernel-resource-usage remarks into one line per kernel."""
import re
import sys

cur = {}
for line in sys.stdin:
    m = re.search(r"remark: (?:Function Name|\s*Name): (\S+)", line) or re.search(r"Name: (\S+)", line)
    if m:
        if cur:
            print(cur)
        cur = {"kernel": m.group(1)[:60]}
        continue
    for key in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize", "Occupancy", "LDS Size"):
        m = re.search(rf"{re.escape(key)}[^:]*: (\d+)", line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
if cur:
    print(cur)
