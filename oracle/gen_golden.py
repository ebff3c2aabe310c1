#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own code (oracle/_ref/libbrickref.so).

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference to have been compiled
by `make -C oracle ref`):      python oracle/gen_golden.py
The fixtures are data only: seeded inputs (or their seeds) and the outputs the reference headers
produced for them.  tests/test_oracle_golden.py replays them against oracle/liboracle.so without
needing the reference.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[0] = ROOT  # not the script dir: `oracle` must resolve to the package
from oracle.oracle import (Oracle, Ref, ST_7PT_COEFF, ST_27PT_COEFF, ST_MPI7, ST_MPI13, ST_MPI25, ST_MPI125,  # noqa: E402
                           TAPS_7PT, taps_27pt)

OUT = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    os.makedirs(OUT, exist_ok=True)
    O, R = Oracle(), Ref()
    manifest = {}

    def save(name, **arrs):
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **arrs)
        manifest[name] = {k: sha(v) for k, v in arrs.items()}

    # ---- input generator samples (pins orc_hash_u01 itself; produced by the oracle, data only)
    idx = np.array([0, 1, 2, 3, 1000, 123456789, 2**40 + 7], dtype=np.uint64)
    save("inputs", idx=idx, seed=np.array([42], dtype=np.uint64),
         val=np.array([O.L.orc_hash_u01(42, int(i)) for i in idx]), coeff129=O.coeffs(129, 1))

    # ---- bitset / neighbours / skin order / zmort
    sets = [[1], [-1], [1, -3], [1, 2, -3], [-1, -2, -3], [2, 3], [3], []]
    zr = np.random.default_rng(7)
    zq = np.stack([zr.integers(0, 1 << 21, 256), zr.integers(0, 3, 256), zr.integers(0, 128, 256)], 1).astype(np.uint64)
    save("bitset_zmort",
         sets_flat=np.array([x for s in sets for x in (s + [0] * (3 - len(s)))], dtype=np.int32),
         sets_val=np.array([R.bitset_of(s) for s in sets], dtype=np.uint64),
         sets_neg=np.array([R.bitset_neg(R.bitset_of(s)) for s in sets], dtype=np.uint64),
         allneighbors=R.allneighbors3(), skin3d_good=R.skin3d_good(), zq=zq,
         zget=np.array([R.zmort_get(int(z), int(d)) for z, d, p in zq], dtype=np.uint64),
         zset=np.array([R.zmort_set(int(z), int(d), int(p)) for z, d, p in zq], dtype=np.uint64))

    # ---- init_grid
    for dims in [(4, 4, 4), (6, 5, 4)]:
        g, a = R.init_grid(dims)
        save("init_grid_%dx%dx%d" % dims, dims=np.array(dims), grid=g, adj=a)

    # ---- single/ shape (stencils/3axis.cpp:14-92,165-247) at N=16: copyToBrick, 7-pt, 27-pt, compareBrick
    N = 16
    STRIDE, STRIDEG = N + 32, N + 16
    SB = STRIDEG // 8
    grid, adj = R.init_grid((SB, SB, SB))
    arr = O.fill_array_global((STRIDE,) * 3, (0, 0, 0), (STRIDE,) * 3, 42)
    coeff = O.coeffs(129, 1)
    nb = SB ** 3
    step = 1024  # bIn at offset 0, bOut at offset 512 (3axis.cpp:25-28)
    st = np.zeros(nb * step)
    R.copy_to_brick((STRIDEG,) * 3, (8,) * 3, (0,) * 3, arr, grid, adj, st, step, 0)
    ids = np.array([grid[(tk * SB + tj) * SB + ti] for tk in range(1, SB - 1) for tj in range(1, SB - 1)
                    for ti in range(1, SB - 1)], dtype=np.uint32)
    bin_ = np.ascontiguousarray(st.reshape(nb, step)[:, :512]).reshape(-1)
    out7 = np.zeros(nb * 512)
    out27 = np.zeros(nb * 512)
    R.brick_taps(adj, bin_, out7, 512, ids, TAPS_7PT, coeff)
    R.brick_taps(adj, bin_, out27, 512, ids, taps_27pt(), coeff)
    # compareBrick verdicts of the reference on an array copy of out7, exact / perturbed
    arr7 = np.zeros_like(arr)
    R.copy_from_brick((N,) * 3, (8,) * 3, (8,) * 3, arr7, grid, adj, out7, 512, 0)
    verdicts = []
    for eps in (0.0, 1e-7, 1e-5):
        a2 = arr7.copy()
        a2[16 + 3, 16 + 2, 16 + 1] += eps
        verdicts.append(int(R.compare_brick((N,) * 3, (8,) * 3, (8,) * 3, a2, grid, adj, out7, 512, 0)))
    save("single_N16", N=np.array([N]), seed=np.array([42, 1], dtype=np.uint64), ids=ids,
         bin_interleaved_sum=np.array([st.sum()]), bin_sample=st.reshape(nb, step)[ids[0], :512].copy(),
         out7=out7.reshape(nb, 512)[ids].copy(), out27=out27.reshape(nb, 512)[ids].copy(),
         arr7_sha=np.frombuffer(bytes.fromhex(sha(arr7)), dtype=np.uint8),
         cmp_eps=np.array([0.0, 1e-7, 1e-5]), cmp_verdict=np.array(verdicts, dtype=np.int32))

    # ---- d3cond (stencils/3axis.cpp:94-163) on a sign-changing field: input = U[0,1) - 0.5
    binc = bin_ - 0.5
    outc = np.zeros(nb * 512)
    R.brick_cond(adj, binc, outc, 512, ids, coeff)
    save("cond_N16", seed=np.array([42, 1], dtype=np.uint64), shift=np.array([-0.5]), ids=ids,
         out=outc.reshape(nb, 512)[ids].copy())

    # ---- accessor samples (brick.h:214-327)
    rng = np.random.default_rng(3)
    q = np.stack([rng.integers(0, nb, 512), rng.integers(-8, 16, 512), rng.integers(-8, 16, 512),
                  rng.integers(-8, 16, 512)], 1).astype(np.int64)
    save("accessor_N16", q=q, val=np.array([R.brick_at(adj, st, step, int(b), int(k), int(j), int(i)) for b, k, j, i in q]))

    # ---- fake.h ST_CPU array sweeps (weak/main.cpp:116-131) on a 40^3 array, interior 24^3
    ext = (40, 40, 40)
    arr = O.fill_array_global(ext, (0, 0, 0), ext, 43)
    lo, hi = (8, 8, 8), (32, 32, 32)
    outs = {}
    for name in ("mpi7", "mpi13", "mpi25", "mpi125"):
        o = np.zeros_like(arr)
        R.array_stencil(name, arr, o, lo, hi)
        outs[name] = o[8:32, 8:32, 8:32].copy()
        outs["iter_" + name] = np.array([R.st_iter(name)])
    save("array_stencils_40", seed=np.array([43], dtype=np.uint64), **outs)

    # ---- BrickDecomp tables (brick-mpi.h:322-460) for several shapes
    for dims in [(16, 16, 16), (32, 32, 32), (24, 32, 40), (64, 64, 64)]:
        d = R.decomp(dims)
        save("decomp_%dx%dx%d" % dims, dims=np.array(dims), sep_pos=np.array(d.sep_pos), nbricks=np.array([d.nbricks]),
             grid=d.grid, adj_from1=d.adj[1:].copy(), skin_size=d.skin_size, ghost=d.ghost, skin=d.skin)
    for dims in [(128, 128, 128), (512, 512, 512)]:
        d = R.decomp(dims)  # too big to store: digests only
        save("decomp_%dx%dx%d" % dims, dims=np.array(dims), sep_pos=np.array(d.sep_pos), nbricks=np.array([d.nbricks]),
             grid_sha=np.frombuffer(bytes.fromhex(sha(d.grid)), dtype=np.uint8),
             adj_from1_sha=np.frombuffer(bytes.fromhex(sha(d.adj[1:])), dtype=np.uint8),
             skin_size=d.skin_size, ghost=d.ghost, skin=d.skin)

    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", len(manifest), "fixtures to", OUT)


if __name__ == "__main__":
    main()
