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


@pytest.mark.parametrize("body,msg", [
    ('f = Func("max", 2)\ng = Grid("a", 3); h = Grid("b", 3)\nh(i, j, k).assign(f(g(i, j, k), ConstRef("0.0")))\nSTENCIL = [h]', "not a linear"),
    ('g = Grid("a", 3); h = Grid("b", 3)\nh(i, j, k).assign(g(i, j, k) * g(i + 1, j, k))\nSTENCIL = [h]', "not a linear"),
    ('g = Grid("a", 3); h = Grid("b", 3)\nh(i, j, k).assign(If(g(i, j, k) > 0, g(i, j, k), -g(i, j, k)))\nSTENCIL = [h]', "not a linear"),
    ('g = Grid("a", 3); h = Grid("b", 3)\nh(i + 1, j, k).assign(g(i, j, k))\nSTENCIL = [h]', "output must be written"),
    ('l = Index(3)\ng = Grid("a", 4); h = Grid("b", 4)\nh(i, j, k, l).assign(g(i, j, k, l))\nSTENCIL = [h]', "only 3-D"),
    ('g = Grid("a", 3); h = Grid("b", 3)\nh(i, j, k).assign(g(j, i, k))\nSTENCIL = [h]', "must be Index"),
])
def test_unsupported_scripts_are_reported(body, msg):
    with pytest.raises(ValueError, match=msg):
        load_spec(HEAD + "from st.func import Func\nfrom st.expr import If\n" + body)


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
