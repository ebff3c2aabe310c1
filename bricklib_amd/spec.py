"""Stencil scripts as data.

The reference describes a stencil in a small Python script (stencils/7pt.py, mpi13pt.py, ...):
    i, j, k = Index(0), Index(1), Index(2);  input = Grid("bIn", 3);  output = Grid("bOut", 3)
    calc = ConstRef("coeff[0]") * input(i, j, k) + ConstRef("coeff[1]") * input(i + 1, j, k) + ...
    output(i, j, k).assign(calc);  STENCIL = [output]
and its code generator turns the expression into vector code (codegen/vecscatter, codegen/st/*).  Here the script is
only *read*: it is executed against the recording stand-ins below for `st.expr` / `st.grid`, which reduce the expression
to a list of taps (offset, coefficient expression).  Linear scripts then run through the generic kernel
(brk_stencil_apply_taps / brklib::brick_apply_taps) or -- when they are one of the built-in stencils -- through the tuned
kernels; anything non-linear (Func, If: cond.py) is reported as such.
This module is written from the script syntax alone; it shares no code with the reference's `st` package."""
import re
import sys
import types


class _Lin:
    """sum of (coefficient, offset) terms; coefficient = product of numbers and ConstRef names"""

    def __init__(self, terms):
        self.terms = terms          # [(num: float, names: tuple[str], offset: tuple[int] or None)]

    @staticmethod
    def lift(x):
        if isinstance(x, _Lin):
            return x
        if isinstance(x, (int, float)):
            return _Lin([(float(x), (), None)])
        raise TypeError(f"unsupported operand {x!r}")

    def __add__(self, o):
        return _Lin(self.terms + _Lin.lift(o).terms)

    __radd__ = __add__

    def __neg__(self):
        return _Lin([(-c, n, o) for c, n, o in self.terms])

    def __sub__(self, o):
        return self + (-_Lin.lift(o))

    def __rsub__(self, o):
        return _Lin.lift(o) + (-self)

    def __mul__(self, o):
        o = _Lin.lift(o)
        out = []
        for c1, n1, o1 in self.terms:
            for c2, n2, o2 in o.terms:
                if o1 is not None and o2 is not None:
                    raise ValueError("product of two grid references: not a linear stencil")
                out.append((c1 * c2, n1 + n2, o1 if o1 is not None else o2))
        return _Lin(out)

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = _Lin.lift(o)
        if len(o.terms) != 1 or o.terms[0][1] or o.terms[0][2] is not None:
            raise ValueError("division by a non-numeric expression")
        return self * (1.0 / o.terms[0][0])

    def __gt__(self, o):
        raise ValueError("comparison in a stencil expression: not a linear stencil")

    __lt__ = __ge__ = __le__ = __gt__


class Index:
    def __init__(self, axis, shift=0):
        self.axis, self.shift = int(axis), int(shift)

    def __add__(self, n):
        return Index(self.axis, self.shift + int(n))

    def __sub__(self, n):
        return Index(self.axis, self.shift - int(n))


def ConstRef(text):
    text = str(text).strip()
    try:
        return _Lin([(float(text), (), None)])
    except ValueError:
        return _Lin([(1.0, (text,), None)])


class _GridRef(_Lin):
    def __init__(self, grid, offset):
        super().__init__([(1.0, (), offset)])
        self.grid, self.offset = grid, offset

    def assign(self, expr):
        if any(self.offset):
            raise ValueError("output must be written at (i, j, k)")
        self.grid.assigned = _Lin.lift(expr)


class Grid:
    def __init__(self, name, dims):
        self.name, self.dims, self.assigned = name, int(dims), None

    def __call__(self, *idx):
        if len(idx) != self.dims:
            raise ValueError(f"grid {self.name}: {len(idx)} indices for {self.dims} dimensions")
        off = [0] * self.dims
        for pos, ix in enumerate(idx):
            if not isinstance(ix, Index) or ix.axis != pos:
                raise ValueError(f"grid {self.name}: argument {pos} must be Index({pos}) +/- constant")
            off[pos] = ix.shift
        g = _GridRef(self, tuple(off))
        g.read_of = self
        return g


def _nonlinear(*a, **k):
    raise ValueError("Func/If in a stencil expression: not a linear stencil")


class Func:
    def __init__(self, *a, **k):
        pass

    __call__ = _nonlinear


If = _nonlinear


class Spec:
    """input/output grid names and the taps of a linear 3-D stencil script"""

    def __init__(self, out_name, terms, grids):
        self.output = out_name
        self._terms = terms
        reads = [t for t in terms if t[2] is not None]
        if len(reads) != len(terms):
            raise ValueError("constant term without a grid reference")
        self.input = grids[0] if grids else None
        self.offsets = [t[2] for t in terms]
        self.radius = max([max(abs(x) for x in o) for o in self.offsets] + [0])

    def coefficient_exprs(self):
        """C expressions of the coefficients, tap by tap (what a rewritten brick(...) line evaluates)"""
        out = []
        for c, names, _ in self._terms:
            parts = ([repr(c)] if c != 1.0 or not names else []) + [f"({n})" for n in names]
            out.append(" * ".join(parts))
        return out

    def taps(self, constants=None):
        """[((di,dj,dk), weight)] with ConstRef names resolved through `constants`: a dict name -> value; entries
        such as "coeff" may be sequences, so "coeff[3]" resolves to constants["coeff"][3]"""
        constants = constants or {}

        def value(name):
            if name in constants:
                return float(constants[name])
            m = re.fullmatch(r"(\w+)\[(\d+)\]", name)
            if m and m.group(1) in constants:
                return float(constants[m.group(1)][int(m.group(2))])
            raise KeyError(f"no value for constant {name!r}")

        out = []
        for c, names, off in self._terms:
            w = c
            for n in names:
                w *= value(n)
            out.append((tuple(off), w))
        return out

    def builtin(self):
        """name of the built-in stencil with exactly these offsets and coefficient expressions, if any"""
        key = sorted(zip(self.offsets, self.coefficient_exprs()))
        for name, want in _BUILTIN.items():
            if key == want:
                return name
        return None


def _star(r, cname):
    t = [((0, 0, 0), cname(0))]
    for d in range(1, r + 1):
        for a in range(3):
            for s in (1, -1):
                o = [0, 0, 0]
                o[a] = s * d
                t.append((tuple(o), cname(d)))
    return sorted(t)


def _cube125():
    t = []
    for dk in range(-2, 3):
        for dj in range(-2, 3):
            for di in range(-2, 3):
                lo, mid, hi = sorted((abs(di), abs(dj), abs(dk)))
                cls = {(0, 0, 0): 0, (0, 0, 1): 1, (0, 0, 2): 2, (0, 1, 1): 3, (0, 1, 2): 4, (0, 2, 2): 5, (1, 1, 1): 6,
                       (1, 1, 2): 7, (1, 2, 2): 8, (2, 2, 2): 9}[(lo, mid, hi)]
                t.append(((di, dj, dk), f"(MPI_C{cls})"))
    return sorted(t)


_BUILTIN = {
    "mpi125pt": _cube125(),
    "mpi7pt": _star(1, lambda d: ("(MPI_ALPHA)", "(MPI_BETA)")[d]),
    "mpi13pt": _star(2, lambda d: f"(MPI_B{d})"),
    "mpi25pt": _star(4, lambda d: f"(MPI_A{d})"),
    "7pt": sorted([((0, 0, 0), "(coeff[0])"), ((1, 0, 0), "(coeff[1])"), ((-1, 0, 0), "(coeff[2])"), ((0, 1, 0), "(coeff[3])"),
                   ((0, -1, 0), "(coeff[4])"), ((0, 0, 1), "(coeff[5])"), ((0, 0, -1), "(coeff[6])")]),
}


def load_spec(source, filename="<stencil script>"):
    """source: the text of a stencil script (or a path to one).  Returns a Spec."""
    if "\n" not in source and source.endswith(".py"):
        filename = source
        source = open(source).read()
    expr = types.ModuleType("st.expr")
    expr.Index, expr.ConstRef, expr.If = Index, ConstRef, If
    grid = types.ModuleType("st.grid")
    made = []

    def grid_factory(name, dims):
        g = Grid(name, dims)
        made.append(g)
        return g

    grid.Grid = grid_factory
    func = types.ModuleType("st.func")
    func.Func = Func
    pkg = types.ModuleType("st")
    pkg.expr, pkg.grid, pkg.func = expr, grid, func
    saved = {k: sys.modules.get(k) for k in ("st", "st.expr", "st.grid", "st.func")}
    sys.modules.update({"st": pkg, "st.expr": expr, "st.grid": grid, "st.func": func})
    try:
        env = {"__name__": "__stencil__"}
        exec(compile(source, filename, "exec"), env)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    outs = env.get("STENCIL")
    if not outs or len(outs) != 1 or outs[0].assigned is None:
        raise ValueError("script must end with STENCIL = [output] after output(i, j, k).assign(expr)")
    out = outs[0]
    if out.dims != 3:
        raise ValueError(f"{out.dims}-D stencil: only 3-D bricks are supported")
    ins = [g.name for g in made if g is not out]
    return Spec(out.name, out.assigned.terms, ins)
