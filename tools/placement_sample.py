#!/usr/bin/env python3
"""How often does brk_storage_alloc find both kinds?  One process = one sample: a 512^3 input storage (plain), then the output storage
placed against it; prints the report (candidates tried, spacers, probe sweep of the chosen / the worst candidate / one allocation).
Run N processes in a row:  for i in $(seq 12); do python tools/placement_sample.py; done"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bricklib_amd as bl  # noqa: E402

dev = torch.device("cuda:0")
n = 66 ** 3
a = bl.BrickStorage.allocate(n, 512, dev)
b = bl.BrickStorage.allocate(n, 512, dev, partners=[(a, "opposite")])
p = b.placement
gain = 1.0 - p["probe_sweep_ms_chosen"] / p["probe_sweep_ms_same_allocation"] if p.get("probe_sweep_ms_same_allocation") else float("nan")
print("candidates %2d chosen %2d spacers %2d GiB  chosen %.4f worst %.4f same-allocation %.4f ms  -> %s" %
      (p["candidates"], p["chosen"], p["spacer_gib"], p["probe_sweep_ms_chosen"], p["probe_sweep_ms_worst"], p["probe_sweep_ms_same_allocation"],
       "both kinds seen" if gain > 0.01 else "ONE KIND ONLY"), flush=True)
