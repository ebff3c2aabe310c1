"""Seeded random stencil scripts in the reference's script syntax (test infrastructure): an expression DAG over grid references
within `radius`, constants and ConstRefs, built from + - * neg max min abs If and comparisons, with sub-expressions used several times.
No division and no square root: every operation is exact-rounded the same way on the host and on the device, so the checks that use
these scripts compare bit for bit."""
import numpy as np

HEAD = """
from st.expr import Index, ConstRef, If
from st.grid import Grid
from st.func import Func
i = Index(0); j = Index(1); k = Index(2)
u = Grid("u", 3); v = Grid("v", 3)
hi = Func("max", 2); lo = Func("fmin", 2); mag = Func("fabs", 1)
"""


def random_script(seed, radius=2, steps=14, ntaps=6):
    rng = np.random.default_rng(seed)
    lines = [HEAD]
    vals = []

    def new(text):
        name = f"t{len(vals)}"
        lines.append(f"{name} = {text}")
        vals.append(name)
        return name

    def idx(ax, d):
        return ax if d == 0 else f"{ax} {'+' if d > 0 else '-'} {abs(d)}"
    new("u(i, j, k)")
    for _ in range(ntaps - 1):
        di, dj, dk = (int(x) for x in rng.integers(-radius, radius + 1, 3))
        new(f"u({idx('i', di)}, {idx('j', dj)}, {idx('k', dk)})")
    new('ConstRef("cA")')
    new(f"ConstRef(\"{rng.uniform(-1, 1):.6f}\")")
    pick = lambda: vals[int(rng.integers(0, len(vals)))]
    recent = lambda: vals[int(rng.integers(max(0, len(vals) - 4), len(vals)))]     # keeps the DAG connected and the live set small
    for _ in range(steps):
        op = int(rng.integers(0, 9))
        a, b, c = recent(), pick(), pick()
        if op == 0:
            new(f"{a} + {b}")
        elif op == 1:
            new(f"{a} - {b}")
        elif op == 2:
            new(f"{a} * {b}")
        elif op == 3:
            new(f"hi({a}, {b})")
        elif op == 4:
            new(f"lo({a}, {b})")
        elif op == 5:
            new(f"mag({a})")
        elif op == 6:
            cmp_ = [">", ">=", "<", "<=", "==", "!="][int(rng.integers(0, 6))]
            new(f"If({a} {cmp_} {b}, {c}, -{a})")
        elif op == 7:
            new(f"{rng.uniform(-2, 2):.4f} * {a} + {b}")
        else:
            new(f"-{a}")
    # the result draws on the last few values so that most of the DAG is live
    lines.append(f"v(i, j, k).assign({vals[-1]} + {vals[-2]} * {vals[-3]} - hi({vals[-4]}, {vals[0]}))")
    lines.append("STENCIL = [v]")
    return "\n".join(lines)
