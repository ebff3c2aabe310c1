#!/usr/bin/env python3
"""Generate tests/golden/{populate_ranks,strong_getrank,dims_create,weak_cycle_ref,strong_cycle_ref}.npz from the REFERENCE run under MPI.

TEST INFRASTRUCTURE ONLY.  Needs the build container: /root/reference, conda MPICH (mpiexec) and
`make -C oracle refmpi` (oracle/_ref/ref_mpi_tables = the reference's populate() from include/brick-mpi.h:730-753,
its strong/args.cpp getrank()/partition compiled as they are, and MPI_Dims_create as weak/args.cpp:105 calls it).
The fixtures are numbers only; tests/test_host_logic.py and tests/test_oracle_golden.py replay them without MPI.
    python oracle/gen_golden_mpi.py
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_mpi_tables")
OUT = os.path.join(ROOT, "tests", "golden")
MPIEXEC = os.environ.get("MPIEXEC", "/opt/conda/bin/mpiexec")
ENV = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib", OMP_NUM_THREADS="1")


def run(np_, *args):
    out = subprocess.run([MPIEXEC, "-np", str(np_), EXE] + [str(a) for a in args], env=ENV, check=True, stdout=subprocess.PIPE,
                         timeout=300).stdout.decode()
    return [l.split() for l in out.splitlines()]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "refmpi"])
    arrs = {}
    # ---- populate(): rank maps of every rank of periodic MPI_Dims_create(N) grids
    sizes = [1, 2, 3, 4, 6, 8, 12, 16]
    arrs_p = {"sizes": np.array(sizes, dtype=np.int32)}
    for n in sizes:
        rows = [l for l in run(n, "populate") if l and l[0] == "P"]
        assert len(rows) == n
        rows.sort(key=lambda l: int(l[2]))
        arrs_p[f"dims_{n}"] = np.array([int(x) for x in rows[0][4:7]], dtype=np.int32)
        arrs_p[f"coo_{n}"] = np.array([[int(x) for x in l[8:11]] for l in rows], dtype=np.int32)
        arrs_p[f"map_{n}"] = np.array([[int(x) for x in l[12:39]] for l in rows], dtype=np.int32)
    # ---- strong: ownership ranges and getrank() of every (subdomain, neighbour)
    cases = [(1, 64, 16), (2, 64, 16), (3, 64, 16), (4, 64, 16), (8, 64, 16), (3, 64, 32), (5, 128, 16), (8, 128, 16), (7, 32, 16)]
    arrs_s = {"cases": np.array(cases, dtype=np.int32)}
    for n, d, s in cases:
        lines = run(n, "strong", "-d", d, "-s", s)
        rng = sorted([(int(l[2]), int(l[8]), int(l[9])) for l in lines if l and l[0] == "S"])
        assert len(rng) == n
        g = np.array([[int(x) for x in l[1:7]] for l in lines if l and l[0] == "G"], dtype=np.uint64)   # rank, z, set, z', dst, sub
        g = g[np.lexsort((g[:, 2], g[:, 1]))]
        assert g.shape[0] == (d // s) ** 3 * 27
        arrs_s[f"range_{n}_{d}_{s}"] = np.array([[a, b] for _, a, b in rng], dtype=np.uint64)
        arrs_s[f"getrank_{n}_{d}_{s}"] = g
    # ---- MPI_Dims_create(n, 3)
    dm = np.array([[int(x) for x in l[1:5]] for l in run(1, "dims") if l and l[0] == "D"], dtype=np.int32)
    # ---- one arr_func of the reference's weak driver (exchangeArr + ST_ITER sweeps of ST_CPU) on 1, 2 and 8 ranks, 16^3 per rank:
    # pins the multi-rank composition (neighbour orientation of populate, ghost exchange, sweep schedule with the skipped
    # outer ring) that the weak-cycle tests reproduce with the oracle on a global periodic array
    import tempfile
    arrs_w = {"seed": np.array([42], dtype=np.uint64), "cells": np.array([16], dtype=np.int32)}
    for st in (7, 13, 25, 125):
        exe = os.path.join(ROOT, "oracle", "_ref", f"ref_mpi_weak_{st}")
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "refweak"])
        for n in (1, 2, 8):
            with tempfile.TemporaryDirectory() as td:
                subprocess.run([MPIEXEC, "-np", str(n), exe, "16", "42", os.path.join(td, "w")], env=ENV, check=True, timeout=600)
                for r in range(n):
                    raw = open(os.path.join(td, f"w_np{n}_r{r}.bin"), "rb").read()
                    hdr = np.frombuffer(raw[:64], dtype=np.int64)
                    assert hdr[0] == n and hdr[1] == r
                    arrs_w[f"hdr_{st}_{n}_{r}"] = hdr.copy()
                    arrs_w[f"out_{st}_{n}_{r}"] = np.frombuffer(raw[64:], dtype=np.float64).reshape(16, 16, 16).copy()
    # ---- one arr_func of the reference's strong driver (its -v validation path) -- 13-point and 7-point -- on 1, 3 and 8 ranks over
    # 32^3 in 16^3 subdomains (all 8 subdomains kept), and on 5 ranks over 64^3 in 16^3 subdomains (every 7th of the 64 kept)
    arrs_g = {"seed": np.array([21], dtype=np.uint64)}
    for st in (13, 7):
        exe = os.path.join(ROOT, "oracle", "_ref", f"ref_mpi_strong_{st}")
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "refstrong"])
        for n, d, sd, keep in ((1, 32, 16, 1), (3, 32, 16, 1), (8, 32, 16, 1), (5, 64, 16, 7)):
            if st == 7 and (n, d) not in ((3, 32), (5, 64)):
                continue
            with tempfile.TemporaryDirectory() as td:
                subprocess.run([MPIEXEC, "-np", str(n), exe, "-d", str(d), "-s", str(sd)], env=dict(ENV, REF_OUT_PREFIX=os.path.join(td, "s")),
                               check=True, timeout=600, stdout=subprocess.DEVNULL)
                nsub = (d // sd) ** 3
                for zid in range(0, nsub, keep):
                    raw = open(os.path.join(td, f"s_np{n}_z{zid}.bin"), "rb").read()
                    hdr = np.frombuffer(raw[:64], dtype=np.int64)
                    assert hdr[0] == n and hdr[2] == zid and hdr[6] == d and hdr[7] == sd
                    arrs_g[f"hdr_{st}_{n}_{d}_{sd}_{zid}"] = hdr.copy()
                    arrs_g[f"out_{st}_{n}_{d}_{sd}_{zid}"] = np.frombuffer(raw[64:], dtype=np.float64).reshape(sd, sd, sd).copy()
    man_path = os.path.join(OUT, "MANIFEST.json")
    manifest = json.load(open(man_path)) if os.path.exists(man_path) else {}
    for name, a in (("populate_ranks", arrs_p), ("strong_getrank", arrs_s), ("dims_create", {"n_dims": dm}), ("weak_cycle_ref", arrs_w), ("strong_cycle_ref", arrs_g)):
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **a)
        manifest[name] = {k: sha(v) for k, v in a.items()}
        print(name, {k: v.shape for k, v in list(a.items())[:4]}, "...")
    json.dump(manifest, open(man_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
