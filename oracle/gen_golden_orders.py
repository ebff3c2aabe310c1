"""Golden BrickDecomp tables for the reference's OTHER surface-section orders (skin3d_normal, skin3d_bad of
src/brick-mpi.cpp:54-136), produced by the reference's own BrickDecomp<3,8,8,8>::initialize compiled in place
(oracle/_ref/libbrickref.so, `make -C oracle ref`).  TEST INFRASTRUCTURE ONLY: writes data (numbers) into tests/golden/ and
merges their SHA-256 into MANIFEST.json; the fixtures of gen_golden.py are left untouched.

    python oracle/gen_golden_orders.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.oracle import Ref  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    R = Ref()
    manifest = json.load(open(os.path.join(OUT, "MANIFEST.json")))
    for order in ("normal", "bad"):
        skin = R.skin3d(order)
        for dims in [(16, 16, 16), (24, 32, 40), (64, 64, 64)]:
            d = R.decomp(dims, order=order)
            name = "decomp_%s_%dx%dx%d" % ((order,) + dims)
            arrs = dict(dims=np.array(dims), skinlist=skin, sep_pos=np.array(d.sep_pos), nbricks=np.array([d.nbricks]), grid=d.grid,
                        adj_from1=d.adj[1:].copy(), skin_size=d.skin_size, ghost=d.ghost, skin=d.skin)
            np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
            manifest[name] = {k: sha(v) for k, v in arrs.items()}
            print(name, "sep", d.sep_pos, "runs", d.nruns, "sections", len(skin))
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
