"""Stencil scripts read as data (bricklib_amd/spec.py): tap extraction, built-in recognition, errors."""
import pytest

from bricklib_amd.spec import load_spec

HEAD = """
from st.expr import Index, ConstRef
from st.grid import Grid
i = Index(0); j = Index(1); k = Index(2)
"""


def star_script(radius, names, gin="in", gout="out"):
    lines = [HEAD, f'src = Grid("{gin}", 3)', f'dst = Grid("{gout}", 3)', f'c = [ConstRef(n) for n in {names!r}]',
             "acc = c[0] * src(i, j, k)"]
    for d in range(1, radius + 1):
        for ax in "ijk":
            for sgn in "+-":
                args = ", ".join(f"{a} {sgn} {d}" if a == ax else a for a in "ijk")
                lines.append(f"acc = acc + c[{d}] * src({args})")
    lines += ["dst(i, j, k).assign(acc)", "STENCIL = [dst]"]
    return "\n".join(lines)


def test_star_scripts_are_recognised_as_builtins():
    s = load_spec(star_script(2, ["MPI_B0", "MPI_B1", "MPI_B2"]))
    assert (s.input, s.output, s.radius, len(s.offsets)) == ("in", "out", 2, 13)
    assert s.builtin() == "mpi13pt"
    assert load_spec(star_script(1, ["MPI_ALPHA", "MPI_BETA"])).builtin() == "mpi7pt"
    assert load_spec(star_script(4, ["MPI_A%d" % q for q in range(5)])).builtin() == "mpi25pt"
    assert load_spec(star_script(2, ["x", "y", "z"])).builtin() is None          # same shape, other constants
    taps = s.taps({"MPI_B0": 0.4, "MPI_B1": 0.07, "MPI_B2": 0.03})
    assert taps[0] == ((0, 0, 0), 0.4) and sorted(w for _, w in taps)[:6] == [0.03] * 6


def test_general_linear_expression():
    s = load_spec(HEAD + """
u = Grid("u", 3); v = Grid("v", 3)
w = [ConstRef("w[%d]" % q) for q in range(3)]
half = ConstRef("0.5")
e = w[0] * u(i, j, k) + half * w[1] * (u(i + 1, j, k) + u(i - 1, j, k)) - w[2] * u(i, j + 2, k - 1) + 2 * u(i, j, k + 1)
v(i, j, k).assign(e / 4)
STENCIL = [v]
""")
    assert s.offsets == [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 2, -1), (0, 0, 1)] and s.radius == 2
    assert s.taps({"w": [1.0, 2.0, 3.0]}) == [((0, 0, 0), 0.25), ((1, 0, 0), 0.25), ((-1, 0, 0), 0.25), ((0, 2, -1), -0.75),
                                              ((0, 0, 1), 0.5)]
    assert s.coefficient_exprs()[3] == "-0.25 * (w[2])"
    with pytest.raises(KeyError):
        s.taps({})


NONLINEAR = [
    'f = Func("max", 2)\ng = Grid("a", 3); h = Grid("b", 3)\nh(i, j, k).assign(f(g(i, j, k), ConstRef("0.0")))\nSTENCIL = [h]',
    'g = Grid("a", 3); h = Grid("b", 3)\nh(i, j, k).assign(g(i, j, k) * g(i + 1, j, k))\nSTENCIL = [h]',
    'g = Grid("a", 3); h = Grid("b", 3)\nh(i, j, k).assign(If(g(i, j, k) > 0, g(i, j, k), -g(i, j, k)))\nSTENCIL = [h]',
    'g = Grid("a", 3); h = Grid("b", 3)\nm = Func("fmin", 2); r = Func("sqrt", 1); a = Func("fabs", 1)\n'
    'e = r(a(g(i, j - 2, k) - g(i, j, k + 2))) / (1 + m(g(i, j, k), g(i - 1, j, k)) * g(i, j, k))\n'
    'h(i, j, k).assign(If(e <= ConstRef("lim"), e, ConstRef("lim")))\nSTENCIL = [h]',
]
PRELUDE = HEAD + "from st.func import Func\nfrom st.expr import If\n"


def run_program(ops, result, tap_value):
    """the register program of Spec.program, interpreted the way expr_brick does (include/brick-device.h)"""
    import numpy as np
    from bricklib_amd.spec import EX
    name = {v: k for k, v in EX.items()}
    reg = {}
    for o in ops:
        x, y, z = (reg.get(o[q]) for q in "abc")
        k = name[o["code"]]
        with np.errstate(all="ignore"):
            reg[o["dst"]] = (tap_value(o["offset"]) if k == "tap" else np.float64(o["imm"]) if k == "const" else
                             x + y if k == "add" else x - y if k == "sub" else x * y if k == "mul" else x / y if k == "div" else
                             -x if k == "neg" else np.where(x > y, x, y) if k == "max" else np.where(x < y, x, y) if k == "min" else
                             np.where(x < 0, -x, x) if k == "abs" else np.sqrt(x) if k == "sqrt" else
                             (x > y) * 1.0 if k == "gt" else (x >= y) * 1.0 if k == "ge" else (x < y) * 1.0 if k == "lt" else
                             (x <= y) * 1.0 if k == "le" else (x == y) * 1.0 if k == "eq" else (x != y) * 1.0 if k == "ne" else
                             np.where(x != 0, y, z))
    return reg[result]


@pytest.mark.parametrize("body", NONLINEAR)
def test_nonlinear_scripts_become_register_programs(body):
    """Func / If / products of grid references (what stencils/cond.py is made of): not a tap list, but a register program for
    brk_stencil_apply_expr whose interpretation equals the tree's numpy evaluation bit for bit"""
    import numpy as np
    from bricklib_amd.spec import MAX_OPS, MAX_REGS
    s = load_spec(PRELUDE + body)
    assert s.linear is False and s.radius <= 2 and s.builtin() is None
    for linear_only in (s.taps, s.coefficient_exprs):
        with pytest.raises(ValueError, match="not a linear"):
            linear_only()
    ops, result = s.program({"lim": 0.75})
    assert 0 < len(ops) <= MAX_OPS and all(0 <= o[q] < MAX_REGS for o in ops for q in ("dst", "a", "b", "c"))
    rng = np.random.default_rng(2)
    vals = {}

    def tap(off):
        return vals.setdefault(tuple(off), rng.random(4096) - 0.5)
    want = s.evaluate(tap, {"lim": 0.75})
    got = run_program(ops, result, tap)
    assert np.array_equal(got, want) and np.isfinite(want).all() and np.unique(want).size > 100
    text, _ = s.program(as_c=True)
    assert all(isinstance(o["imm"], (float, int, str)) for o in text)


def test_linear_scripts_also_compile_to_programs():
    import numpy as np
    s = load_spec(star_script(2, ["MPI_B0", "MPI_B1", "MPI_B2"]))
    consts = {"MPI_B0": 0.4, "MPI_B1": 0.07, "MPI_B2": 0.03}
    ops, result = s.program(consts)
    rng = np.random.default_rng(3)
    vals = {}

    def tap(off):
        return vals.setdefault(tuple(off), rng.random(512))
    assert s.linear and np.array_equal(run_program(ops, result, tap), s.evaluate(tap, consts))
    assert np.allclose(s.evaluate(tap, consts), sum(w * tap(o) for o, w in s.taps(consts)), rtol=1e-14)


def test_expression_limits_are_reported():
    wide = "g = Grid('a', 3); h = Grid('b', 3)\nf = Func('max', 2)\nh(i, j, k).assign(f(g(i + %d, j, k), g(i, j, k)))\nSTENCIL = [h]"
    with pytest.raises(ValueError, match="<= 4 per axis"):
        load_spec(PRELUDE + wide % 5).program()
    assert load_spec(PRELUDE + wide % 4).program()[0][0]["offset"] in ((4, 0, 0), (0, 0, 0))      # radius 3 and 4: the larger tiles
    many = "g = Grid('a', 3); h = Grid('b', 3)\nf = Func('max', 2)\ne = g(i, j, k)\n" + \
        "for q in range(70):\n    e = f(e, g(i + q % 3 - 1, j, k)) * g(i, j + q % 5 - 2, k)\n" + "h(i, j, k).assign(e)\nSTENCIL = [h]"
    with pytest.raises(ValueError, match="operations"):
        load_spec(PRELUDE + many).program()
    with pytest.raises(ValueError, match="unknown function|Func"):
        load_spec(PRELUDE + "g = Grid('a', 3); h = Grid('b', 3)\nf = Func('erf', 1)\nh(i, j, k).assign(f(g(i, j, k)))\nSTENCIL = [h]")


@pytest.mark.parametrize("body,msg", [
    ('g = Grid("a", 3); h = Grid("b", 3)\nh(i + 1, j, k).assign(g(i, j, k))\nSTENCIL = [h]', "output must be written"),
    ('l = Index(3)\ng = Grid("a", 4); h = Grid("b", 4)\nh(i, j, k, l).assign(g(i, j, k, l))\nSTENCIL = [h]', "only 3-D"),
    ('g = Grid("a", 3); h = Grid("b", 3)\nh(i, j, k).assign(g(j, i, k))\nSTENCIL = [h]', "must be Index"),
])
def test_unsupported_scripts_are_reported(body, msg):
    with pytest.raises(ValueError, match=msg):
        load_spec(PRELUDE + body)


COND_LIKE = PRELUDE + """
pos = Func("max", 2)
u = Grid("in", 3); v = Grid("out", 3)
c = [ConstRef("coeff[%d]" % q) for q in range(7)]
zero = ConstRef("0.0")
total = None
for q, (di, dj, dk) in ((5, (0, 0, 1)), (6, (0, 0, -1)), (3, (0, 1, 0)), (4, (0, -1, 0)), (1, (1, 0, 0)), (2, (-1, 0, 0)), (0, (0, 0, 0))):
    term = c[q] * pos(u(i + di, j + dj, k + dk), zero)
    total = term if total is None else total + term
v(i, j, k).assign(If(total > 0, total, -total))
STENCIL = [v]
"""


def test_oracle_expression_evaluation_agrees_with_the_pinned_conditional_stencil(oracle):
    """oracle.brick_expr (the checker of the expression path) on a script of the shape of the reference's stencils/cond.py, against the
    oracle's own d3cond restatement, which tests/test_oracle_golden.py pins against the reference's header: bit for bit (same
    operations in the same order; coefficients of both signs so that both arms of the If are taken)"""
    import numpy as np
    from oracle.oracle import ST_COND7
    grid, adj = oracle.init_grid((5, 5, 5))
    rng = np.random.default_rng(7)
    hin = rng.random(adj.shape[0] * 512) - 0.5
    g3 = grid.reshape(5, 5, 5)
    ids = np.array([g3[2, 2, 2], g3[2, 1, 3], g3[3, 3, 1]], dtype=np.uint32)
    coeff = oracle.coeffs(129, 1) - 0.45
    s = load_spec(COND_LIKE)
    a, b = np.zeros_like(hin), np.zeros_like(hin)
    oracle.brick_stencil(ST_COND7, adj, hin, a, 512, ids=ids, coeff=coeff)
    oracle.brick_expr(s, adj, hin, b, 512, ids, {"coeff": coeff})
    assert np.array_equal(a, b) and np.count_nonzero(b) > 1400     # (a cell whose seven taps are all negative gives 0)
    total = np.zeros_like(hin)
    oracle.brick_taps(adj, np.maximum(hin, 0), total, 512, ids, [(o, coeff[q]) for q, o in
                      ((5, (0, 0, 1)), (6, (0, 0, -1)), (3, (0, 1, 0)), (4, (0, -1, 0)), (1, (1, 0, 0)), (2, (-1, 0, 0)), (0, (0, 0, 0)))])
    assert (total < 0).any() and (total > 0).any()      # both arms of the If


def test_oracle_tap_evaluation_agrees_with_pinned_stencils(oracle):
    """orc_brick_taps (the checker of the generic tap path) against the oracle's own 7- and 13-point brick stencils, which
    are pinned bit-exactly against the reference (tests/test_oracle_golden.py): same numbers up to the order of additions."""
    import numpy as np
    from oracle.oracle import ST_7PT_COEFF, ST_MPI13
    dims = (5, 5, 5)
    grid, adj = oracle.init_grid(dims)
    rng = np.random.default_rng(4)
    hin = rng.random(adj.shape[0] * 512)
    ids = np.array([grid.reshape(5, 5, 5)[2, 2, 2], grid.reshape(5, 5, 5)[2, 1, 3]], dtype=np.uint32)
    coeff = oracle.coeffs(129, 1)
    seven = [((0, 0, 0), coeff[0]), ((1, 0, 0), coeff[1]), ((-1, 0, 0), coeff[2]), ((0, 1, 0), coeff[3]), ((0, -1, 0), coeff[4]),
             ((0, 0, 1), coeff[5]), ((0, 0, -1), coeff[6])]
    s = load_spec(star_script(2, ["MPI_B0", "MPI_B1", "MPI_B2"]))
    thirteen = s.taps({"MPI_B0": 0.4, "MPI_B1": 0.07, "MPI_B2": 0.03})
    for sid, taps, c in ((ST_7PT_COEFF, seven, coeff), (ST_MPI13, thirteen, None)):
        a, b = np.zeros_like(hin), np.zeros_like(hin)
        oracle.brick_stencil(sid, adj, hin, a, 512, ids=ids, coeff=c)
        oracle.brick_taps(adj, hin, b, 512, ids, taps)
        assert np.abs(a - b).max() <= 1e-14 and np.abs(b).max() > 0


def test_expression_source_generation_and_compilation_without_a_device():
    """brk_expr_source: the translation unit of a program (one statement per operation, constants as exact hexadecimal literals,
    taps as tile offsets).  brk_expr_compile gets as far as a code object for gfx950 here -- hiprtc needs no GPU -- and then reports
    that there is no device to load it on (BRK_ENODEV); with a device: tests/test_gpu_parity.py."""
    import torch
    from bricklib_amd._lib import BrkError
    from bricklib_amd.stencils import CompiledExpr, expr_source
    s = load_spec(PRELUDE + NONLINEAR[3])
    text = expr_source(s, {"lim": 0.75})
    ops, result = s.program({"lim": 0.75})
    assert text.count(";\n    r") + 1 >= len(ops) and f"return r{result};" in text
    # every grid reference of this script moves along one axis (radius 2): the cell update of the star kernel k_star_wg, not the cube tile
    assert "brk_user_star(const double *ck" in text and "jm[1]" in text and "ck[2]" in text and "im[0]" in text and "0x1.8p-1" in text
    assert "template __global__ void k_star_wg<kStarUser2, 4, 4>(" in text
    assert "__builtin_sqrt(" in text
    cube = expr_source(load_spec(PRELUDE + NONLINEAR[1].replace("g(i + 1, j, k)", "g(i + 1, j - 2, k)")))
    # taps off the axes, radius 2: the cell update of the cube kernel (plane pitch 12 rows x 8)
    assert "brk_user_cube(const double *c, const int *col)" in cube and "c[-16 + col[1]]" in cube and "k_cube_dense<kCubeUser2, 4, 3, 5>(" in cube
    far = expr_source(load_spec(PRELUDE + NONLINEAR[1].replace("g(i + 1, j, k)", "g(i + 3, j - 2, k)")))
    assert "brklib::brick_apply_fn<3>(" in far and "c[brklib::tap_rel<3>(3, -2, 0)]" in far      # beyond radius 2: the (8 + 2R)^3 tile
    assert expr_source(s, {"lim": 0.75}) == text and expr_source(s, {"lim": 0.5}) != text
    # a star-shaped program of radius 1: the 7-point kernel itself, instantiated around the program as its cell update
    seven = expr_source(load_spec(COND_LIKE), {"coeff": [0.1] * 7})
    assert "brk_user_point7(double kp, double km, double jp, double jm, double ip, double im, double c0)" in seven
    assert "template __global__ void k_s7_wg<kUser, 8, kModeList>(" in seven and "r0 = kp;" in seven
    if torch.cuda.is_available():
        return
    # every form of generated unit compiles here and defines the kernel its launcher will ask for by name (checked before the device is):
    # star radius 2 (k_star_wg), star radius 1 (k_s7_wg, list + range), star radius 4, off-axis radius 1 / 2 (k_cube_dense), radius 3 (plain tile)
    off = lambda a, b: load_spec(PRELUDE + NONLINEAR[1].replace("g(i + 1, j, k)", "g(i + %d, j - %d, k)" % (a, b)))
    for sp, consts in ((s, {"lim": 0.75}), (load_spec(COND_LIKE), {"coeff": [0.1] * 7}), (off(4, 0), None), (off(1, 1), None), (off(1, 2), None), (off(3, 2), None)):
        with pytest.raises(BrkError, match="compiled for gfx950 .* no device|libhiprtc"):
            CompiledExpr(sp, consts)


@pytest.mark.parametrize("steps", [14, 45])
def test_random_scripts_compile_to_equivalent_programs(steps):
    """seeded random expression DAGs (tests/random_scripts.py: shared sub-expressions, If, comparisons, max / min / abs, radius 1..4):
    the register program interpreted on the host equals the tree's evaluation bit for bit; the few that need more than 16 live values
    or 120 operations say so"""
    import numpy as np
    from random_scripts import random_script
    from bricklib_amd.spec import MAX_OPS, MAX_REGS
    from bricklib_amd.stencils import expr_source
    compiled = refused = 0
    for seed in range(40):
        s = load_spec(random_script(seed, radius=1 + seed % 4, steps=steps))
        try:
            ops, result = s.program({"cA": 0.3})
        except ValueError as e:
            assert "values at once" in str(e) or "operations" in str(e)
            refused += 1
            continue
        assert len(ops) <= MAX_OPS and all(o[q] < MAX_REGS for o in ops for q in ("dst", "a", "b", "c"))
        rng = np.random.default_rng(seed)
        vals = {}

        def tap(off):
            return vals.setdefault(tuple(off), rng.random(1500) - 0.5)
        assert np.array_equal(run_program(ops, result, tap), s.evaluate(tap, {"cA": 0.3})), seed
        if seed % 8 == 0:
            assert f"return r{result};" in expr_source(s, {"cA": 0.3})
        compiled += 1
    assert compiled >= 30, (compiled, refused)


def test_compiled_form_takes_longer_programs_than_the_interpreter():
    """the interpreter's program is a kernel argument (120 operations); the compiled form is text (512)"""
    from bricklib_amd.spec import MAX_OPS, MAX_OPS_COMPILED
    from bricklib_amd.stencils import expr_source
    body = "g = Grid('a', 3); h = Grid('b', 3)\nf = Func('max', 2)\ne = g(i, j, k)\n" + \
        "for q in range(%d):\n    e = f(e, g(i + q %% 3 - 1, j, k)) * g(i, j + q %% 5 - 2, k)\n" + "h(i, j, k).assign(e)\nSTENCIL = [h]"
    s = load_spec(PRELUDE + body % 70)
    with pytest.raises(ValueError, match="at most %d" % MAX_OPS):
        s.program()
    ops, _ = s.program(max_ops=MAX_OPS_COMPILED)
    assert MAX_OPS < len(ops) <= MAX_OPS_COMPILED and "brk_user_star(" in expr_source(s)
    with pytest.raises(ValueError, match="at most %d" % MAX_OPS_COMPILED):
        load_spec(PRELUDE + body % 200).program(max_ops=MAX_OPS_COMPILED)
