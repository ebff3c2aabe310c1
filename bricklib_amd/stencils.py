"""Stencil apply -- the hot path.  Python face of brk_stencil_apply_{range,list} (include/brk.h)."""
import ctypes as C

import numpy as np

from ._lib import (BRK_7PT_COEFF, BRK_27PT_COEFF, BRK_MPI7, BRK_MPI13, BRK_MPI25, BRK_MPI125, BRK_SWEEP_KPAIRED, BRK_SWEEP_REVERSE, BrkError,
                   check, lib)
from .brick import _ptr, _stream

# name -> id; names follow the reference's stencil scripts (stencils/*.py) and drivers
STENCILS = {"7pt": BRK_7PT_COEFF, "27pt": BRK_27PT_COEFF, "mpi7pt": BRK_MPI7, "mpi13pt": BRK_MPI13,
            "mpi25pt": BRK_MPI25, "mpi125pt": BRK_MPI125, "cond": 6,
            # two consecutive 7-point sweeps in one pass (not in the reference; see include/brk.h)
            "mpi7pt_x2": 7, "7pt_x2": 8}


def _sid(stencil):
    return STENCILS[stencil] if isinstance(stencil, str) else int(stencil)


def stencil_radius(stencil):
    return lib.brk_stencil_radius(_sid(stencil))


def stencil_iters(stencil):
    """ST_ITER of stencils/fake.h: sweeps per ghost exchange."""
    return lib.brk_stencil_iters(_sid(stencil))


def apply_stencil(stencil, bIn, bOut, ids=None, brick_range=None, box=None, coeff=None, stream=None, kpaired=False, reverse=False):
    """out[b] = stencil(in[b], neighbours) for b in `ids` (device int32 tensor holding uint32 brick ids, e.g.
    the reference's `grid`), in brick_range=(begin,end), or in box=(base, tot[3], lo[3], cnt[3]) of a lexicographic
    numbering (contiguous axis first).  bIn/bOut: Brick objects sharing one BrickInfo.  kpaired: `ids` is in the k-paired
    order of bricklib_amd.order.tiled(..., kpair=True) (BRK_SWEEP_KPAIRED: faster for the 25-point star, same results).
    reverse: walk the same bricks back to front (BRK_SWEEP_REVERSE); alternate it from one sweep to the next and the sweep starts
    where the last one stopped -- in the part of the field the Infinity Cache still holds (class Zigzag below)."""
    sid = _sid(stencil)
    cp = None
    if coeff is not None:
        cbuf = np.ascontiguousarray(coeff, dtype=np.float64)
        cp = cbuf.ctypes.data_as(C.POINTER(C.c_double))
    adj = bIn.bInfo.adj
    flags = (BRK_SWEEP_KPAIRED if kpaired and ids is not None and box is None else 0) | (BRK_SWEEP_REVERSE if reverse else 0)
    if box is not None:
        base, tot, lo, cnt = box
        U3 = C.c_uint32 * 3
        check(lib.brk_stencil_apply_box_ex(sid, _ptr(adj), _ptr(bIn.dat), _ptr(bOut.dat), bIn.step, int(base),
                                           U3(*[int(x) for x in tot]), U3(*[int(x) for x in lo]), U3(*[int(x) for x in cnt]),
                                           cp, flags, _stream(stream)))
    elif ids is not None:
        check(lib.brk_stencil_apply_list_ex(sid, _ptr(adj), _ptr(bIn.dat), _ptr(bOut.dat), bIn.step, _ptr(ids), ids.numel(), cp, flags,
                                            _stream(stream)))
    else:
        b0, b1 = brick_range if brick_range is not None else (0, bIn.bInfo.nbricks)
        check(lib.brk_stencil_apply_range_ex(sid, _ptr(adj), _ptr(bIn.dat), _ptr(bOut.dat), bIn.step, int(b0), int(b1), cp, flags,
                                             _stream(stream)))


def _expr_table(spec, constants, max_ops=None):
    from ._lib import ExprOp
    from .spec import MAX_OPS
    ops, result = spec.program(constants, max_ops=max_ops or MAX_OPS)
    tab = (ExprOp * len(ops))()
    for q, o in enumerate(ops):
        tab[q].code, tab[q].dst, tab[q].a, tab[q].b, tab[q].c, tab[q].imm = o["code"], o["dst"], o["a"], o["b"], o["c"], float(o["imm"])
        tab[q].offset[0], tab[q].offset[1], tab[q].offset[2] = o["offset"]
    return tab, len(ops), result


class CompiledExpr:
    """A stencil script compiled at run time for the current device (brk_expr_compile: the script's operations as straight-line HIP
    over the staged tile, through hiprtc; about a second, once per distinct program).  The counterpart of the reference's generate-then-
    compile step (codegen/vecscatter) for a host that holds the script as data.  apply(bIn, bOut, ids= | brick_range=); .source is the
    generated translation unit."""

    def __init__(self, spec, constants=None):
        import ctypes as C
        from .spec import MAX_OPS_COMPILED
        tab, n, result = _expr_table(spec, constants, MAX_OPS_COMPILED)
        self._h = C.c_void_p()
        check(lib.brk_expr_compile(tab, n, result, C.byref(self._h)))

    @property
    def source(self):
        return lib.brk_expr_kernel_source(self._h).decode()

    @property
    def from_disk(self):
        """True when the code object came out of BRK_JIT_CACHE_DIR instead of the compiler"""
        return bool(lib.brk_expr_kernel_from_disk(self._h))

    def apply(self, bIn, bOut, ids=None, brick_range=None, box=None, stream=None):
        adj = bIn.bInfo.adj
        if box is not None:      # (base, tot[3], lo[3], cnt[3]) of a lexicographic numbering, as for apply_stencil
            base, tot, lo, cnt = box
            u3 = lambda v: (C.c_uint32 * 3)(*[int(x) for x in v])
            check(lib.brk_expr_apply_box(self._h, _ptr(adj), _ptr(bIn.dat), _ptr(bOut.dat), bIn.step, int(base), u3(tot), u3(lo), u3(cnt), _stream(stream)))
        elif ids is not None:
            check(lib.brk_expr_apply(self._h, _ptr(adj), _ptr(bIn.dat), _ptr(bOut.dat), bIn.step, _ptr(ids), 0, ids.numel(), _stream(stream)))
        else:
            b0, b1 = brick_range if brick_range is not None else (0, bIn.bInfo.nbricks)
            check(lib.brk_expr_apply(self._h, _ptr(adj), _ptr(bIn.dat), _ptr(bOut.dat), bIn.step, None, int(b0), int(b1 - b0), _stream(stream)))

    def release(self):
        if self._h:
            lib.brk_expr_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def expr_source(spec, constants=None):
    """the translation unit brk_expr_compile would compile for this script (no compiler, no device needed)"""
    import ctypes as C
    from .spec import MAX_OPS_COMPILED
    tab, n, result = _expr_table(spec, constants, MAX_OPS_COMPILED)
    need = C.c_size_t()
    check(lib.brk_expr_source(tab, n, result, None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    check(lib.brk_expr_source(tab, n, result, buf, need.value, None))
    return buf.value.decode()


def apply_expr(spec, bIn, bOut, ids=None, brick_range=None, constants=None, stream=None, interpret=False, box=None):
    """A stencil script as an expression: `spec` = bricklib_amd.spec.load_spec(...) of ANY script up to radius 4 -- Func("max" / "min" /
    "abs" / "sqrt"), comparisons, If, products of grid references included (what the reference's generator accepts; stencils/cond.py is
    made of these).  constants: values of the script's ConstRef names.
    Default: the script is COMPILED at run time on first use (CompiledExpr: hiprtc, about a second, kept on `spec` per set of constants)
    and runs as generated code.  interpret=True: brk_stencil_apply_expr, the same operations interpreted -- no compiler involved, about
    10x slower, programs up to 120 operations.  Both give the same numbers bit for bit.  Linear scripts are better served by apply_taps
    / the built-in stencils."""
    if not interpret:
        from .spec import MAX_OPS_COMPILED
        tab, n, result = _expr_table(spec, constants, MAX_OPS_COMPILED)
        cache = spec.__dict__.setdefault("_compiled", {})
        key = (bytes(tab), result)
        ck = cache.get(key)
        if ck is None or not ck._h:
            ck = cache[key] = CompiledExpr(spec, constants)
        try:
            return ck.apply(bIn, bOut, ids=ids, brick_range=brick_range, box=box, stream=stream)
        except BrkError as e:
            if "compile the program again" not in str(e):
                raise
        ck = cache[key] = CompiledExpr(spec, constants)       # unloaded by finalize(): the same kernel, loaded again
        return ck.apply(bIn, bOut, ids=ids, brick_range=brick_range, box=box, stream=stream)
    if box is not None:
        raise ValueError("the interpreter takes brick lists and ranges; box= needs the compiled form")
    tab, n, result = _expr_table(spec, constants)
    adj = bIn.bInfo.adj
    if ids is not None:
        check(lib.brk_stencil_apply_expr(_ptr(adj), _ptr(bIn.dat), _ptr(bOut.dat), bIn.step, _ptr(ids), 0, ids.numel(), tab, n, result, _stream(stream)))
    else:
        b0, b1 = brick_range if brick_range is not None else (0, bIn.bInfo.nbricks)
        check(lib.brk_stencil_apply_expr(_ptr(adj), _ptr(bIn.dat), _ptr(bOut.dat), bIn.step, None, int(b0), int(b1 - b0), tab, n, result, _stream(stream)))


def apply_script(spec, bIn, bOut, ids=None, brick_range=None, constants=None, stream=None, box=None):
    """A stencil script, whatever it is -- the front end's dispatch.  Returns the path taken:
      "builtin:<name>"  the script is one of the reference's stencils (same offsets, same coefficient expressions, the MPI_* constants at
                        their fake.h values): its tuned kernel;
      "compiled"        non-linear scripts, and linear ones of up to 40 taps: compiled at run time on first use (star-shaped ones into the
                        tuned star kernels), apply_expr;
      "taps"            linear scripts with more taps: the tap tables, apply_taps (brick lists and ranges only).
    box=(base, tot[3], lo[3], cnt[3]): the interior box of a lexicographic numbering instead of ids / brick_range, as for apply_stencil."""
    from .spec import MPI_DEFAULTS
    consts = dict(MPI_DEFAULTS)
    consts.update(constants or {})
    name = spec.builtin()
    if name == "7pt" and "coeff" in consts:
        coeff = np.zeros(129)
        c = np.asarray(consts["coeff"], dtype=np.float64)
        coeff[:min(c.size, 129)] = c[:129]
        apply_stencil("7pt", bIn, bOut, ids=ids, brick_range=brick_range, box=box, coeff=coeff, stream=stream)
        return "builtin:7pt"
    if name and name.startswith("mpi") and all(consts[k] == v for k, v in MPI_DEFAULTS.items()):
        apply_stencil(name, bIn, bOut, ids=ids, brick_range=brick_range, box=box, stream=stream)
        return "builtin:" + name
    if spec.linear and len(spec.offsets) > 40:
        if box is not None:
            raise ValueError("the tap tables take brick lists and ranges (bricklib_amd.order.tiled builds the list of a box)")
        apply_taps(spec.taps(consts), bIn, bOut, ids=ids, brick_range=brick_range, stream=stream)
        return "taps"
    apply_expr(spec, bIn, bOut, ids=ids, brick_range=brick_range, constants=consts, stream=stream, box=box)
    return "compiled"


def last_kernel():
    """Name (template arguments included) of the kernel the last stencil launch of the calling thread enqueued (brk_last_kernel)."""
    return lib.brk_last_kernel().decode()


def finalize():
    """brk_finalize: wait for the library's work on the current device and release its device-side caches (before process exit)."""
    check(lib.brk_finalize())


class Zigzag:
    """Direction of successive sweeps over the same bricks: `z.turn()` returns False, True, False, ... -- pass it as `reverse=`."""

    def __init__(self):
        self.reverse = True

    def turn(self):
        self.reverse = not self.reverse
        return self.reverse


def apply_taps(taps, bIn, bOut, ids=None, brick_range=None, stream=None):
    """A linear stencil given as data (brk_stencil_apply_taps): taps = [((di, dj, dk), weight), ...], applied in list order -- any
    table with offsets in [-4, 4] (up to 729 taps; inside the 125-point cube and star-shaped tables have lighter kernels); e.g. the output of
    bricklib_amd.spec.load_spec(...).taps(constants)."""
    off = np.ascontiguousarray([list(o) for o, _ in taps], dtype=np.int32).reshape(-1)
    w = np.ascontiguousarray([float(x) for _, x in taps], dtype=np.float64)
    offp, wp = off.ctypes.data_as(C.POINTER(C.c_int32)), w.ctypes.data_as(C.POINTER(C.c_double))
    adj = bIn.bInfo.adj
    if ids is not None:
        check(lib.brk_stencil_apply_taps(_ptr(adj), _ptr(bIn.dat), _ptr(bOut.dat), bIn.step, _ptr(ids), 0, ids.numel(),
                                         len(taps), offp, wp, _stream(stream)))
    else:
        b0, b1 = brick_range if brick_range is not None else (0, bIn.bInfo.nbricks)
        check(lib.brk_stencil_apply_taps(_ptr(adj), _ptr(bIn.dat), _ptr(bOut.dat), bIn.step, None, int(b0), int(b1 - b0),
                                         len(taps), offp, wp, _stream(stream)))
