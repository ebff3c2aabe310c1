#!/usr/bin/env python3
"""Time the expression path (brk_stencil_apply_expr, k_expr) on a 512^3 field: a cond-shaped script beside the tuned BRK_COND7 kernel
and the generic tap path, a radius-2 limiter script, the minmod example.  ms per sweep, CUDA-event timed, 20 sweeps after 3 warm-ups."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bricklib_amd as bl  # noqa: E402
from bricklib_amd.spec import load_spec  # noqa: E402

HEAD = """
from st.expr import Index, ConstRef, If
from st.grid import Grid
from st.func import Func
i = Index(0); j = Index(1); k = Index(2)
u = Grid("u", 3); v = Grid("v", 3)
"""
COND = HEAD + """
pos = Func("max", 2)
total = None
for q, (di, dj, dk) in ((5, (0, 0, 1)), (6, (0, 0, -1)), (3, (0, 1, 0)), (4, (0, -1, 0)), (1, (1, 0, 0)), (2, (-1, 0, 0)), (0, (0, 0, 0))):
    term = ConstRef("coeff[%d]" % q) * pos(u(i + di, j + dj, k + dk), 0.0)
    total = term if total is None else total + term
v(i, j, k).assign(If(total > 0, total, -total))
STENCIL = [v]
"""
LIMITER = HEAD + """
lo = Func("min", 2); mag = Func("fabs", 1)
dl = u(i, j, k) - u(i - 1, j, k); dr = u(i + 1, j, k) - u(i, j, k)
far = u(i + 2, j - 1, k) * u(i - 2, j, k + 2) - u(i, j + 2, k - 2)
v(i, j, k).assign(u(i, j, k) + 0.3 * If(dl * dr > 0, lo(mag(dl), mag(dr)), 0.0) * If(far >= 0, 1.0, -1.0))
STENCIL = [v]
"""


def main():
    dev = torch.device("cuda:0")
    n = 512 // 8 + 2
    grid, info = bl.init_grid((n, n, n), dev)
    a, b = (bl.BrickStorage.allocate(info.nbricks, 512, dev) for _ in range(2))
    a.dat.uniform_(-0.5, 0.5)
    A, B = bl.Brick(info, a, 0), bl.Brick(info, b, 0)
    from bricklib_amd.order import tiled
    from bricklib_amd.brick import grid_to_device
    ids = grid_to_device(tiled(grid.reshape(n, n, n)[1:-1, 1:-1, 1:-1]), dev)
    coeff = [0.1 * (q - 3) for q in range(7)]
    minmod = load_spec(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples", "minmod.py"))
    cond = load_spec(COND)

    def timed(name, fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:44s} {ms:7.3f} ms/sweep  {512 ** 3 / ms / 1e6:7.1f} GStencil/s  {bl.last_kernel()}", flush=True)

    timed("cond.py tuned kernel (BRK_COND7)", lambda: bl.apply_stencil("cond", A, B, ids=ids, coeff=coeff + [0.0] * 122))
    timed("cond-shaped script, interpreted", lambda: bl.apply_expr(cond, A, B, ids=ids, constants={"coeff": coeff}, interpret=True))
    import time
    t0 = time.time()
    ck = bl.CompiledExpr(cond, {"coeff": coeff})
    print(f"  (brk_expr_compile: {time.time() - t0:.2f} s)")
    timed("cond-shaped script, compiled at run time", lambda: ck.apply(A, B, ids=ids))
    lin = load_spec(COND.replace('pos(u(i + di, j + dj, k + dk), 0.0)', 'u(i + di, j + dj, k + dk)').replace("If(total > 0, total, -total)", "total"))
    taps = lin.taps({"coeff": coeff})
    timed("same 7 taps, linear tap path (k_taps)", lambda: bl.apply_taps(taps, A, B, ids=ids))
    timed("same 7 taps, linear, interpreted", lambda: bl.apply_expr(lin, A, B, ids=ids, constants={"coeff": coeff}, interpret=True))
    cl = bl.CompiledExpr(lin, {"coeff": coeff})
    timed("same 7 taps, linear, compiled", lambda: cl.apply(A, B, ids=ids))
    timed("minmod.py (3 limited slopes), interpreted", lambda: bl.apply_expr(minmod, A, B, ids=ids, constants={"coeff": coeff}, interpret=True))
    cm = bl.CompiledExpr(minmod, {"coeff": coeff})
    timed("minmod.py (3 limited slopes), compiled", lambda: cm.apply(A, B, ids=ids))
    lim = load_spec(LIMITER)
    timed("radius-2 limiter, interpreted", lambda: bl.apply_expr(lim, A, B, ids=ids, interpret=True))
    c2 = bl.CompiledExpr(lim)
    timed("radius-2 limiter, compiled", lambda: c2.apply(A, B, ids=ids))
    lap = load_spec(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples", "lap19.py"))
    cl19 = bl.CompiledExpr(lap, {"coeff": coeff})
    timed("lap19.py (19 taps, faces + edges), compiled", lambda: cl19.apply(A, B, ids=ids))
    t19 = lap.taps({"coeff": coeff})
    timed("lap19.py, tap path (k_taps)", lambda: bl.apply_taps(t19, A, B, ids=ids))
    mixed = load_spec(HEAD + """
hi = Func("max", 2)
uxy = u(i + 1, j + 1, k) - u(i + 1, j - 1, k) - u(i - 1, j + 1, k) + u(i - 1, j - 1, k)
uyz = u(i, j + 1, k + 1) - u(i, j + 1, k - 1) - u(i, j - 1, k + 1) + u(i, j - 1, k - 1)
v(i, j, k).assign(u(i, j, k) + 0.25 * hi(uxy, uyz) * If(uxy * uyz > 0, 1.0, 0.5))
STENCIL = [v]
""")
    cmx = bl.CompiledExpr(mixed)
    timed("mixed derivatives (radius 1, off-axis, max / If), compiled", lambda: cmx.apply(A, B, ids=ids))
    timed("built-in 27-point kernel, same list", lambda: bl.apply_stencil("27pt", A, B, ids=ids, coeff=[0.01 * q for q in range(129)]))
    s2 = load_spec(HEAD + """
lo = Func("min", 2); mag = Func("fabs", 1)
d4 = lambda a, b, c, d: (8.0 * (c - b) - (d - a)) / 12.0
gx = d4(u(i - 2, j, k), u(i - 1, j, k), u(i + 1, j, k), u(i + 2, j, k))
gy = d4(u(i, j - 2, k), u(i, j - 1, k), u(i, j + 1, k), u(i, j + 2, k))
gz = d4(u(i, j, k - 2), u(i, j, k - 1), u(i, j, k + 1), u(i, j, k + 2))
v(i, j, k).assign(u(i, j, k) - 0.3 * (lo(mag(gx), mag(gy)) + If(gz > 0, gz, 0.5 * gz)))
STENCIL = [v]
""")
    cs2 = bl.CompiledExpr(s2)
    timed("radius-2 STAR limiter (4th-order slopes), compiled", lambda: cs2.apply(A, B, ids=ids))
    timed("built-in 13-point star, same list", lambda: bl.apply_stencil("mpi13pt", A, B, ids=ids))
    l13 = load_spec(HEAD + "acc = 0.3 * u(i, j, k)\nfor d in (1, 2):\n  for s in (1, -1):\n    acc = acc + (0.1 / d) * u(i + s * d, j, k) + (0.08 / d) * u(i, j + s * d, k) + (0.06 / d) * u(i, j, k + s * d)\nv(i, j, k).assign(acc)\nSTENCIL = [v]\n")
    c13 = bl.CompiledExpr(l13)
    timed("13-point star with its own weights, compiled", lambda: c13.apply(A, B, ids=ids))
    t13 = l13.taps()
    timed("the same 13 taps, tap path (k_star_taps)", lambda: bl.apply_taps(t13, A, B, ids=ids))
    # beyond the 125-point cube: (8 + 2R)^3 tiles, 2 / 1 waves per workgroup
    import itertools
    for r in (2, 3, 4):
        cube = [((di, dj, dk), 1.0 / (1 + abs(di) + abs(dj) + abs(dk))) for dk, dj, di in itertools.product(range(-r, r + 1), repeat=3)]
        timed(f"all {len(cube)} taps of the radius-{r} cube (tap path)", lambda: bl.apply_taps(cube, A, B, ids=ids))
    # linear scripts with many taps, compiled (straight-line code, weights as literals) against the table paths
    for r in (2,):       # (radius 3: 1 371 operations, beyond the compiled form's limit -- it spilled: 87 ms)
        text = HEAD + "acc = None\nfor dk in range(-%d, %d):\n  for dj in range(-%d, %d):\n    for di in range(-%d, %d):\n      t = (1.0 / (1 + abs(di) + abs(dj) + abs(dk))) * u(i + di, j + dj, k + dk)\n      acc = t if acc is None else acc + t\nv(i, j, k).assign(acc)\nSTENCIL = [v]\n" % (r, r + 1, r, r + 1, r, r + 1)
        sp = load_spec(text)
        t0 = time.time()
        cc = bl.CompiledExpr(sp)
        print(f"  (brk_expr_compile, {(2 * r + 1) ** 3} taps: {time.time() - t0:.2f} s)")
        timed(f"all {(2 * r + 1) ** 3} taps of the radius-{r} cube, compiled", lambda: cc.apply(A, B, ids=ids))
    far = load_spec(HEAD + """
hi = Func("max", 2)
v(i, j, k).assign(hi(u(i + 3, j - 3, k + 1), u(i - 2, j + 3, k - 3)) * u(i, j, k) - u(i - 3, j, k))
STENCIL = [v]
""")
    timed("radius-3 off-axis max/product, interpreted", lambda: bl.apply_expr(far, A, B, ids=ids, interpret=True))
    c3 = bl.CompiledExpr(far)
    timed("radius-3 off-axis max/product, compiled", lambda: c3.apply(A, B, ids=ids))
    far4 = load_spec(HEAD + """
v(i, j, k).assign(If(u(i + 4, j - 4, k + 4) > u(i - 4, j + 1, k - 2), u(i, j + 4, k), u(i - 1, j, k) * u(i, j, k)))
STENCIL = [v]
""")
    c4 = bl.CompiledExpr(far4)
    timed("radius-4 off-axis If, compiled", lambda: c4.apply(A, B, ids=ids))


if __name__ == "__main__":
    main()
