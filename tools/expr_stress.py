#!/usr/bin/env python3
"""One-off stress of the stencils-as-data paths (round 5): 160 seeded random scripts (tests/random_scripts.py; radius 1..4, 20..49
combining steps) -- the interpreter and the run-time compiled kernel agree bit for bit on every one, every fourth also with the oracle's
evaluation of the tree.  Result on the GPU box: 160 of 160 in 47.5 s (0.3 s per compile)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bricklib_amd as bl
from bricklib_amd.spec import load_spec
from random_scripts import random_script
from oracle.oracle import Oracle
O = Oracle()
dev = torch.device("cuda:0")
d = bl.BrickDecomp((24, 16, 32)); dinfo = d.getBrickInfo(dev)
A, B, C = (dinfo.allocate(512, dev) for _ in range(3))
A.dat.uniform_(-0.5, 0.5); A.dat[:512] = 0
hin = A.dat.cpu().numpy()
rng = (1, d.sep_pos[2]); ids = np.arange(rng[0], rng[1], dtype=np.uint32)
ok = refused = 0; t0 = time.time()
for seed in range(100, 260):
    spec = load_spec(random_script(seed, radius=1 + seed % 4, steps=20 + seed % 30, ntaps=4 + seed % 7))
    try:
        spec.program({"cA": 0.3})
    except ValueError:
        refused += 1; continue
    bl.apply_expr(spec, bl.Brick(dinfo, A, 0), bl.Brick(dinfo, B, 0), brick_range=rng, constants={"cA": 0.3}, interpret=True)
    ck = bl.CompiledExpr(spec, {"cA": 0.3}); ck.apply(bl.Brick(dinfo, A, 0), bl.Brick(dinfo, C, 0), brick_range=rng); ck.release()
    assert torch.equal(B.dat, C.dat), seed
    if seed % 4 == 0:
        want = np.zeros_like(hin); O.brick_expr(spec, d.adj, hin, want, 512, ids, {"cA": 0.3})
        assert np.array_equal(B.dat.cpu().numpy(), want), seed
    ok += 1
    if seed % 20 == 0: print(seed, ok, refused, round(time.time() - t0, 1), flush=True)
print("random scripts: compiled == interpreted bit for bit:", ok, "refused (too many live values / ops):", refused, "in", round(time.time() - t0, 1), "s")
