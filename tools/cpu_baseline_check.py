#!/usr/bin/env python3
"""BASELINE.md section 4.2, run in the build container: the CPU port that bench.py times as `cpu_baseline` (oracle/brick_oracle.c,
OpenMP over bricks) beside the container figures of the compiled reference (same 8 vCPUs), and the
reference's own accessor path (oracle/_ref/libbrickref.so: include/brick.h _BrickAccessor driven by the 7-point tap table) where
that library is present.  Protocol: time_func (stencils/stencils.h:40-53) with TOT_TIME = 5 s.
    python tools/cpu_baseline_check.py > profiles/r02_e_cpu_baseline_container_check.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bricklib_amd.timing import time_func  # noqa: E402  (pure Python; bricklib_amd/_lib needs libbrk.so, which build() provides)
from oracle.oracle import ST_7PT_COEFF, TAPS_7PT, Oracle, Ref  # noqa: E402

REF_S2 = {256: {"Arr": 0.0173, "Bri": 0.0183, "Trans": 0.00988}, 512: {"Arr": 0.1178, "Bri": 0.1487, "Trans": 0.0787}}  # BASELINE.md s.2, s/sweep


def main():
    O = Oracle()
    threads = len(os.sched_getaffinity(0))
    O.L.orc_set_threads(threads)
    print(f"# container: {threads} threads; protocol time_func, TOT_TIME 5 s; 7-point brick sweep, 7pt.py coefficients")
    try:
        R = Ref()
    except Exception as e:
        R = None
        print("# oracle/_ref/libbrickref.so not available:", e)
    for N in (256, 512):
        SB = (N + 16) // 8
        grid, adj = O.init_grid((SB, SB, SB))
        ids = np.ascontiguousarray(grid.reshape(SB, SB, SB)[1:-1, 1:-1, 1:-1]).reshape(-1).astype(np.uint32)
        rng = np.random.default_rng(1)
        fin = rng.random(adj.shape[0] * 512)
        out = np.zeros_like(fin)
        coeff = rng.random(129)
        O.brick_stencil(ST_7PT_COEFF, adj, fin, out, 512, ids=ids, coeff=coeff)
        sec, calls, dt = time_func(lambda: O.brick_stencil(ST_7PT_COEFF, adj, fin, out, 512, ids=ids, coeff=coeff), 5.0)
        line = f"N={N}: port (oracle/brick_oracle.c) {sec:.5f} s/sweep = {N ** 3 / sec / 1e9:.2f} GStencil/s over {calls} sweeps"
        ref = REF_S2[N]
        line += " | reference binary in this container (BASELINE.md s.2): " + ", ".join(f"{k} {v} s = {N ** 3 / v / 1e9:.2f}" for k, v in ref.items())
        line += f" | port / Trans time = {sec / ref['Trans']:.2f}, port / Arr = {sec / ref['Arr']:.2f}"
        print(line)
        if R is not None and N == 256:
            out2 = np.zeros_like(fin)
            taps = TAPS_7PT
            R.brick_taps(adj, fin, out2, 512, ids, taps, coeff)
            sec2, calls2, _ = time_func(lambda: R.brick_taps(adj, fin, out2, 512, ids, taps, coeff), 5.0)
            print(f"N={N}: reference accessor (include/brick.h compiled in place, naive `Bri` form) {sec2:.5f} s/sweep = {N ** 3 / sec2 / 1e9:.2f} "
                  f"GStencil/s over {calls2} sweeps; max |port - reference| = {np.abs(out - out2)[np.repeat(ids.astype(np.int64) * 512, 512) + np.tile(np.arange(512), ids.size)].max():.3e}")


if __name__ == "__main__":
    main()
