"""Stencil scripts as data.

The reference describes a stencil in a small Python script (stencils/7pt.py, mpi13pt.py, cond.py ...):
    i, j, k = Index(0), Index(1), Index(2);  input = Grid("bIn", 3);  output = Grid("bOut", 3)
    calc = ConstRef("coeff[0]") * input(i, j, k) + ConstRef("coeff[1]") * input(i + 1, j, k) + ...
    output(i, j, k).assign(calc);  STENCIL = [output]
and its code generator turns the expression into vector code (codegen/vecscatter, codegen/st/*).  Here the script is
only *read*: it is executed against the recording stand-ins below for `st.expr` / `st.grid` / `st.func`, which keep the
expression as a tree.  From the tree:
  * a LINEAR script reduces to a list of taps (offset, coefficient expression): one of the built-in stencils -> its tuned kernel;
    anything else up to radius 4 (cube- or star-shaped) -> brk_stencil_apply_taps / brklib::brick_apply_taps;
  * a NON-LINEAR script (Func("max" | "min" | "abs" | "fabs" | "sqrt" ...), comparisons, If, products of grid references: what
    stencils/cond.py is made of; round 5) becomes a small register program (Spec.program) for brk_stencil_apply_expr /
    brklib::brick_apply_fn, or compiled at run time (brk_expr_compile) -- the general path, up to radius 4.
This module is written from the script syntax alone; it shares no code with the reference's `st` package."""
import re
import sys
import types

_FUNCS = {"max": ("max", 2), "fmax": ("max", 2), "min": ("min", 2), "fmin": ("min", 2), "abs": ("abs", 1), "fabs": ("abs", 1),
          "sqrt": ("sqrt", 1)}
_CMP = {">": "gt", ">=": "ge", "<": "lt", "<=": "le", "==": "eq", "!=": "ne"}


class _E:
    """expression node: kind in tap | num | cref | neg | add | sub | mul | div | max | min | abs | sqrt | gt.. | if"""

    def __init__(self, kind, args=(), value=None):
        self.kind, self.args, self.value = kind, tuple(args), value

    @staticmethod
    def lift(x):
        if isinstance(x, _E):
            return x
        if isinstance(x, (int, float)) and not isinstance(x, bool):
            return _E("num", value=float(x))
        raise TypeError(f"unsupported operand {x!r}")

    def _bin(self, kind, o, swap=False):
        o = _E.lift(o)
        return _E(kind, (o, self) if swap else (self, o))

    def __add__(self, o): return self._bin("add", o)
    def __radd__(self, o): return self._bin("add", o, True)
    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, True)
    def __truediv__(self, o): return self._bin("div", o)
    def __rtruediv__(self, o): return self._bin("div", o, True)
    def __neg__(self): return _E("neg", (self,))
    def __gt__(self, o): return self._bin("gt", o)
    def __ge__(self, o): return self._bin("ge", o)
    def __lt__(self, o): return self._bin("lt", o)
    def __le__(self, o): return self._bin("le", o)
    def __eq__(self, o): return self._bin("eq", o)     # noqa: the script's `==` builds a node (nodes are compared by identity here)
    def __ne__(self, o): return self._bin("ne", o)
    __hash__ = object.__hash__

    def __bool__(self):
        raise ValueError("a stencil expression has no truth value in the script: use If(cond, then, else)")


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
        return _E("num", value=float(text))
    except ValueError:
        return _E("cref", value=text)


class _GridRef(_E):
    def __init__(self, grid, offset):
        super().__init__("tap", value=tuple(offset))
        self.grid, self.offset = grid, tuple(offset)

    def assign(self, expr):
        if any(self.offset):
            raise ValueError("output must be written at (i, j, k)")
        self.grid.assigned = _E.lift(expr)


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
        return _GridRef(self, off)


class Func:
    """Func("max", 2): the reference passes the name through to C; here a fixed set is understood"""

    def __init__(self, name, nargs=None):
        self.name = str(name)
        if self.name not in _FUNCS:
            raise ValueError(f'Func("{self.name}"): functions understood in a stencil script: {", ".join(sorted(_FUNCS))}')
        self.kind, self.nargs = _FUNCS[self.name]
        if nargs is not None and int(nargs) != self.nargs:
            raise ValueError(f'Func("{self.name}", {nargs}): takes {self.nargs} argument(s)')

    def __call__(self, *args):
        if len(args) != self.nargs:
            raise ValueError(f"{self.name}: {len(args)} arguments for {self.nargs}")
        return _E(self.kind, [_E.lift(a) for a in args])


def If(cond, then, other):
    return _E("if", (_E.lift(cond), _E.lift(then), _E.lift(other)))


class NotLinear(ValueError):
    pass


def _linear_terms(e):
    """[(number, names, offset or None)] of a linear expression; NotLinear otherwise"""
    k = e.kind
    if k == "tap":
        return [(1.0, (), e.value)]
    if k == "num":
        return [(e.value, (), None)]
    if k == "cref":
        return [(1.0, (e.value,), None)]
    if k == "neg":
        return [(-c, n, o) for c, n, o in _linear_terms(e.args[0])]
    if k in ("add", "sub"):
        a, b = _linear_terms(e.args[0]), _linear_terms(e.args[1])
        return a + ([(-c, n, o) for c, n, o in b] if k == "sub" else b)
    if k == "mul":
        out = []
        for c1, n1, o1 in _linear_terms(e.args[0]):
            for c2, n2, o2 in _linear_terms(e.args[1]):
                if o1 is not None and o2 is not None:
                    raise NotLinear("product of two grid references: not a linear stencil")
                out.append((c1 * c2, n1 + n2, o1 if o1 is not None else o2))
        return out
    if k == "div":
        d = _linear_terms(e.args[1])
        if len(d) != 1 or d[0][1] or d[0][2] is not None:
            raise NotLinear("division by a non-numeric expression: not a linear stencil")
        return [(c / d[0][0], n, o) for c, n, o in _linear_terms(e.args[0])]
    raise NotLinear(f"{k} in a stencil expression: not a linear stencil")


def _resolver(constants):
    constants = constants or {}

    def value(name):
        if name in constants:
            return float(constants[name])
        m = re.fullmatch(r"(\w+)\[(\d+)\]", name)
        if m and m.group(1) in constants:
            return float(constants[m.group(1)][int(m.group(2))])
        raise KeyError(f"no value for constant {name!r}")
    return value


EX = {"tap": 0, "const": 1, "add": 2, "sub": 3, "mul": 4, "div": 5, "neg": 6, "max": 7, "min": 8, "abs": 9, "sqrt": 10, "gt": 11, "ge": 12,
      "lt": 13, "le": 14, "eq": 15, "ne": 16, "if": 17}       # BRK_EX_* of include/brk.h
MAX_REGS, MAX_OPS, MAX_OPS_COMPILED = 16, 120, 512     # interpreter: the program travels as a kernel argument; compiled: as text


class Spec:
    """input/output grid names of a 3-D stencil script; taps when it is linear, a register program in any case"""

    def __init__(self, out_name, expr, grids):
        self.output, self.expr = out_name, expr
        self.input = grids[0] if grids else None
        try:
            self._terms = _linear_terms(expr)
            if any(t[2] is None for t in self._terms):
                raise ValueError("constant term without a grid reference")
            self.linear = True
        except NotLinear as e:
            self._terms, self.linear, self._why = None, False, str(e)
        offs = []
        self._walk(expr, lambda n: offs.append(n.value) if n.kind == "tap" else None, set())
        if not offs:
            raise ValueError("the expression reads no grid")
        self.radius = max(max(abs(x) for x in o) for o in offs)
        self._tap_offsets = offs

    @staticmethod
    def _walk(e, fn, seen):
        if id(e) in seen:
            return
        seen.add(id(e))
        for a in e.args:
            Spec._walk(a, fn, seen)
        fn(e)

    def _need_linear(self):
        if not self.linear:
            raise ValueError(self._why)

    @property
    def offsets(self):
        self._need_linear()
        return [t[2] for t in self._terms]

    def coefficient_exprs(self):
        """C expressions of the coefficients, tap by tap (what a rewritten brick(...) line evaluates)"""
        self._need_linear()
        out = []
        for c, names, _ in self._terms:
            parts = ([repr(c)] if c != 1.0 or not names else []) + [f"({n})" for n in names]
            out.append(" * ".join(parts))
        return out

    def taps(self, constants=None):
        """[((di,dj,dk), weight)] with ConstRef names resolved through `constants`: a dict name -> value; entries
        such as "coeff" may be sequences, so "coeff[3]" resolves to constants["coeff"][3]"""
        self._need_linear()
        value = _resolver(constants)
        out = []
        for c, names, off in self._terms:
            w = c
            for n in names:
                w *= value(n)
            out.append((tuple(off), w))
        return out

    def builtin(self):
        """name of the built-in stencil with exactly these offsets and coefficient expressions, if any"""
        if not self.linear:
            return None
        key = sorted(zip(self.offsets, self.coefficient_exprs()))
        for name, want in _BUILTIN.items():
            if key == want:
                return name
        return None

    def program(self, constants=None, as_c=False, max_ops=MAX_OPS):
        """(ops, result register): the expression as a register program for brk_stencil_apply_expr -- ops are dicts
        {code, dst, a, b, c, offset, imm}; at most 16 registers live at a time (a sub-expression used several times, like `calc` in
        cond.py's If(calc > 0, calc, -calc), is computed once and kept).  constants: as for taps(); as_c: `imm` of a ConstRef stays its
        C text (what tools/brkscatter writes into a driver, where `coeff[3]` is an lvalue of the kernel)."""
        if self.radius > 4:
            raise ValueError(f"radius {self.radius}: stencils given as data reach |offset| <= 4 per axis")
        import sys
        if sys.getrecursionlimit() < 20000:      # a sum of hundreds of taps is a tree that deep
            sys.setrecursionlimit(20000)
        value = _resolver(constants)
        uses = {}

        def count(e):
            uses[id(e)] = uses.get(id(e), 0) + 1
            if uses[id(e)] == 1:
                for a in e.args:
                    count(a)
        count(self.expr)
        need = {}          # Sethi-Ullman number: registers a subtree wants (shared nodes: as if unshared; only an ordering hint)

        def weigh(e):
            if id(e) in need:
                return need[id(e)]
            w = sorted((weigh(a) for a in e.args), reverse=True)
            need[id(e)] = max([1] + [x + q for q, x in enumerate(w)])
            return need[id(e)]
        weigh(self.expr)
        ops, loc, free = [], {}, list(range(MAX_REGS - 1, -1, -1))

        def release(e):
            uses[id(e)] -= 1
            if uses[id(e)] == 0:
                free.append(loc.pop(id(e)))

        def gen(e):
            if id(e) in loc:
                return loc[id(e)]
            order = sorted(range(len(e.args)), key=lambda q: -need[id(e.args[q])])
            regs = {}
            for q in order:
                regs[q] = gen(e.args[q])
            src = [regs[q] for q in range(len(e.args))]
            for a in e.args:            # sources may be reused as the destination
                release(a)
            if not free:
                raise ValueError(f"the expression needs more than {MAX_REGS} values at once")
            dst = free.pop()
            op = {"code": EX["const" if e.kind in ("num", "cref") else e.kind], "dst": dst, "a": 0, "b": 0, "c": 0, "offset": (0, 0, 0), "imm": 0.0}
            if e.kind == "tap":
                op["offset"] = tuple(e.value)
            elif e.kind == "num":
                op["imm"] = e.value
            elif e.kind == "cref":
                op["imm"] = f"({e.value})" if as_c else value(e.value)
            else:
                for name, r in zip("abc", src):
                    op[name] = r
            ops.append(op)
            loc[id(e)] = dst
            return dst

        result = gen(self.expr)
        if len(ops) > max_ops:
            raise ValueError(f"the expression compiles to {len(ops)} operations (at most {max_ops})")
        return ops, result

    def evaluate(self, tap_value, constants=None):
        """numpy evaluation of the expression: tap_value(offset) -> array of in[p + offset] over the cells wanted (the checker's side
        of the expression path: oracle.brick_expr feeds it with the accessor's values)"""
        import sys
        import numpy as np
        if sys.getrecursionlimit() < 20000:
            sys.setrecursionlimit(20000)
        value = _resolver(constants)
        memo = {}

        def ev(e):
            if id(e) in memo:
                return memo[id(e)]
            k, a = e.kind, [ev(x) for x in e.args]
            if k == "tap":
                r = tap_value(tuple(e.value))
            elif k == "num":
                r = e.value
            elif k == "cref":
                r = value(e.value)
            elif k == "neg":
                r = -a[0]
            elif k == "add":
                r = a[0] + a[1]
            elif k == "sub":
                r = a[0] - a[1]
            elif k == "mul":
                r = a[0] * a[1]
            elif k == "div":
                r = a[0] / a[1]
            elif k == "max":
                r = np.where(np.asarray(a[0]) > np.asarray(a[1]), a[0], a[1])
            elif k == "min":
                r = np.where(np.asarray(a[0]) < np.asarray(a[1]), a[0], a[1])
            elif k == "abs":
                r = np.where(np.asarray(a[0]) < 0, -np.asarray(a[0]), a[0])
            elif k == "sqrt":
                r = np.sqrt(a[0])
            elif k in ("gt", "ge", "lt", "le", "eq", "ne"):
                f = {"gt": np.greater, "ge": np.greater_equal, "lt": np.less, "le": np.less_equal, "eq": np.equal, "ne": np.not_equal}[k]
                r = f(a[0], a[1]).astype(np.float64)
            elif k == "if":
                r = np.where(np.asarray(a[0]) != 0, a[1], a[2])
            else:
                raise ValueError(k)
            memo[id(e)] = r
            return r
        return ev(self.expr)


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


# the constants of the reference's MPI stencils (stencils/fake.h:11-33): what the built-in kernels compute with
MPI_DEFAULTS = {"MPI_ALPHA": 0.4, "MPI_BETA": 0.1, "MPI_B0": 0.4, "MPI_B1": 0.07, "MPI_B2": 0.03,
                "MPI_A0": 0.1, "MPI_A1": 0.06, "MPI_A2": 0.045, "MPI_A3": 0.03, "MPI_A4": 0.015,
                **{f"MPI_C{q}": v for q, v in enumerate((0.1, 0.04, 0.03, 0.01, 0.006, 0.004, 0.005, 0.002, 0.003, 0.001))}}


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
    return Spec(out.name, out.assigned, ins)
