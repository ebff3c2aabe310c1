"""ctypes bindings for the CPU checker.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
`Oracle` wraps oracle/liboracle.so (our CPU restatement, brick_oracle.c); `Ref` wraps
oracle/_ref/libbrickref.so (the reference's own headers compiled in place; present only where
`make -C oracle ref` was run, i.e. in the build container, and shipped prebuilt to the GPU box).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

ST_7PT_COEFF, ST_27PT_COEFF, ST_MPI7, ST_MPI13, ST_MPI25, ST_MPI125, ST_COND7 = range(7)
STENCIL_NAMES = {0: "7pt", 1: "27pt", 2: "mpi7", 3: "mpi13", 4: "mpi25", 5: "mpi125", 6: "cond"}

_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def _l3(v):
    return (C.c_long * 3)(*[int(x) for x in v])


def _u3(v):
    return (C.c_uint32 * 3)(*[int(x) for x in v])


def _coeff(c):
    if c is None:
        return None
    c = np.ascontiguousarray(c, dtype=np.float64)
    return c.ctypes.data_as(C.POINTER(C.c_double)), c


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        L = self.L = C.CDLL(path)
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_hash_u01.restype = C.c_double
        L.orc_hash_u01.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_fill_array_global.argtypes = [_f64p, C.c_long * 3, C.c_long * 3, C.c_long * 3, C.c_uint64]
        L.orc_coeffs.argtypes = [_f64p, C.c_int, C.c_uint64]
        L.orc_init_grid.argtypes = [C.c_long * 3, _u32p, _u32p]
        L.orc_brick_at.restype = C.c_double
        L.orc_brick_at.argtypes = [_u32p, _f64p, C.c_size_t, C.c_uint32, C.c_int, C.c_int, C.c_int]
        for f in (L.orc_copy_to_brick, L.orc_copy_from_brick):
            f.argtypes = [C.c_long * 3, C.c_long * 3, C.c_long * 3, _f64p, _u32p, _f64p, C.c_size_t]
        L.orc_compare_brick.restype = C.c_int
        L.orc_compare_brick.argtypes = [C.c_long * 3, C.c_long * 3, C.c_long * 3, _f64p, _u32p, _f64p, C.c_size_t,
                                        C.c_double]
        L.orc_maxdiff_brick.restype = C.c_double
        L.orc_maxdiff_brick.argtypes = [C.c_long * 3, C.c_long * 3, C.c_long * 3, _f64p, _u32p, _f64p, C.c_size_t]
        L.orc_array_stencil.argtypes = [C.c_int, _f64p, _f64p, C.c_long * 3, C.c_long * 3, C.c_long * 3,
                                        C.POINTER(C.c_double)]
        L.orc_brick_stencil.argtypes = [C.c_int, _u32p, _f64p, _f64p, C.c_size_t, C.c_void_p, C.c_long, C.c_uint32,
                                        C.POINTER(C.c_double)]
        L.orc_brick_taps.argtypes = [_u32p, _f64p, _f64p, C.c_size_t, _u32p, C.c_long, C.c_int,
                                     np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS"), _f64p]
        L.orc_brick_taps.restype = None
        L.orc_brick_stencil_naive.argtypes = [C.c_int, _u32p, _f64p, _f64p, C.c_size_t, _u32p, C.c_long,
                                              C.POINTER(C.c_double)]
        L.orc_stencil_radius.argtypes = [C.c_int]
        L.orc_stencil_ncoeff.argtypes = [C.c_int]
        L.orc_bitset_of.restype = C.c_uint64
        L.orc_bitset_of.argtypes = [_i32p, C.c_int]
        L.orc_bitset_neg.restype = C.c_uint64
        L.orc_bitset_neg.argtypes = [C.c_uint64]
        L.orc_allneighbors3.argtypes = [_u64p]
        L.orc_skin3d_good.argtypes = [_u64p]
        L.orc_decomp_build.restype = C.c_void_p
        L.orc_decomp_build.argtypes = [C.c_uint32 * 3, C.c_uint32]
        L.orc_decomp_build_order.restype = C.c_void_p
        L.orc_decomp_build_order.argtypes = [C.c_uint32 * 3, C.c_uint32, _u64p, C.c_int]
        L.orc_skin3d.argtypes = [C.c_int, _u64p]
        L.orc_decomp_free.argtypes = [C.c_void_p]
        L.orc_decomp_info.argtypes = [C.c_void_p, _u32p]
        L.orc_decomp_grid.argtypes = [C.c_void_p, _u32p]
        L.orc_decomp_adj.argtypes = [C.c_void_p, _u32p]
        L.orc_decomp_skin_size.argtypes = [C.c_void_p, _i64p]
        L.orc_decomp_runs.argtypes = [C.c_void_p, C.c_int, _u64p]
        L.orc_populate_rank_map.argtypes = [C.c_int * 3, C.c_int * 3, _i32p]
        L.orc_exchange.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, _i32p, C.c_size_t]
        L.orc_zmort_get.restype = C.c_uint64
        L.orc_zmort_get.argtypes = [C.c_uint64] * 3
        L.orc_zmort_set.restype = C.c_uint64
        L.orc_zmort_set.argtypes = [C.c_uint64] * 4
        L.orc_strong_partition.argtypes = [C.c_uint64, C.c_int, C.c_int, _u64p]
        L.orc_strong_getrank.restype = C.c_uint64
        L.orc_strong_getrank.argtypes = [C.c_uint64] * 6 + [C.POINTER(C.c_int), C.POINTER(C.c_int)]

    # ---- inputs
    def fill_array_global(self, ext, origin, gdims, seed):
        arr = np.empty((ext[2], ext[1], ext[0]), dtype=np.float64)
        self.L.orc_fill_array_global(arr, _l3(ext), _l3(origin), _l3(gdims), seed)
        return arr

    def coeffs(self, n, seed):
        c = np.empty(n, dtype=np.float64)
        self.L.orc_coeffs(c, n, seed)
        return c

    # ---- bricksetup
    def init_grid(self, dims):
        size = int(dims[0]) * int(dims[1]) * int(dims[2])
        grid = np.empty(size, dtype=np.uint32)
        adj = np.empty((size, 27), dtype=np.uint32)
        self.L.orc_init_grid(_l3(dims), grid, adj)
        return grid, adj

    def brick_at(self, adj, dat, step, b, k, j, i):
        return self.L.orc_brick_at(adj, dat, step, b, k, j, i)

    def copy_to_brick(self, dimlist, padding, ghost, arr, grid, bdat, step):
        self.L.orc_copy_to_brick(_l3(dimlist), _l3(padding), _l3(ghost), arr.reshape(-1), grid, bdat, step)

    def copy_from_brick(self, dimlist, padding, ghost, arr, grid, bdat, step):
        self.L.orc_copy_from_brick(_l3(dimlist), _l3(padding), _l3(ghost), arr.reshape(-1), grid, bdat, step)

    def compare_brick(self, dimlist, padding, ghost, arr, grid, bdat, step, tol=1e-6):
        return bool(self.L.orc_compare_brick(_l3(dimlist), _l3(padding), _l3(ghost), arr.reshape(-1), grid, bdat,
                                             step, tol))

    def maxdiff_brick(self, dimlist, padding, ghost, arr, grid, bdat, step):
        return self.L.orc_maxdiff_brick(_l3(dimlist), _l3(padding), _l3(ghost), arr.reshape(-1), grid, bdat, step)

    # ---- stencils
    def radius(self, sid):
        return self.L.orc_stencil_radius(sid)

    def ncoeff(self, sid):
        return self.L.orc_stencil_ncoeff(sid)

    def array_stencil(self, sid, arr_in, arr_out, lo, hi, coeff=None):
        ext = (arr_in.shape[2], arr_in.shape[1], arr_in.shape[0])
        cp = _coeff(coeff)
        self.L.orc_array_stencil(sid, arr_in.reshape(-1), arr_out.reshape(-1), _l3(ext), _l3(lo), _l3(hi),
                                 cp[0] if cp else None)

    def brick_stencil(self, sid, adj, bin_, bout, step, ids=None, n=None, first=0, coeff=None):
        cp = _coeff(coeff)
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.uint32)
            n = ids.size
            idp = ids.ctypes.data_as(C.c_void_p)
        else:
            idp = None
        self.L.orc_brick_stencil(sid, adj, bin_, bout, step, idp, n, first, cp[0] if cp else None)

    def brick_taps(self, adj, bin_, bout, step, ids, taps):
        """taps: [((di,dj,dk), weight), ...]"""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        off = np.ascontiguousarray([o for o, _ in taps], dtype=np.int32).reshape(-1)
        w = np.ascontiguousarray([x for _, x in taps], dtype=np.float64)
        self.L.orc_brick_taps(adj, bin_, bout, step, ids, ids.size, len(taps), off, w)

    def brick_expr(self, spec, adj, bin_, bout, step, ids, constants=None):
        """A stencil script as an EXPRESSION (bricklib_amd.spec.Spec: Func / If / products allowed; what the reference's generator
        accepts, codegen/st/codegen/base.py:678) evaluated cell by cell with the accessor's neighbour semantics: every grid reference
        in[p + offset] is gathered with orc_brick_taps (one tap of weight 1: exactly the value), the tree is evaluated in numpy.
        The checker of brk_stencil_apply_expr."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        cache = {}
        tmp = np.zeros_like(bin_)       # one scratch field: orc_brick_taps writes the listed bricks only

        def tap(off):
            if off not in cache:
                self.brick_taps(adj, bin_, tmp, step, ids, [(off, 1.0)])
                cache[off] = tmp.reshape(-1, step)[ids][:, :512].copy()
            return cache[off]
        val = np.broadcast_to(np.asarray(spec.evaluate(tap, constants), dtype=np.float64), (ids.size, 512))
        bout.reshape(-1, step)[ids, :512] = val

    def brick_stencil_naive(self, sid, adj, bin_, bout, step, ids, coeff=None):
        cp = _coeff(coeff)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        self.L.orc_brick_stencil_naive(sid, adj, bin_, bout, step, ids, ids.size, cp[0] if cp else None)

    # ---- bitset / decomp
    def bitset_of(self, elems):
        e = np.asarray(elems, dtype=np.int32)
        return int(self.L.orc_bitset_of(e, e.size))

    def bitset_neg(self, s):
        return int(self.L.orc_bitset_neg(s))

    def allneighbors3(self):
        out = np.empty(27, dtype=np.uint64)
        self.L.orc_allneighbors3(out)
        return out

    def skin3d_good(self):
        out = np.empty(26, dtype=np.uint64)
        self.L.orc_skin3d_good(out)
        return out

    def decomp(self, dims_cells, ghost_cells=8, order="good"):
        """order: "good" (skin3d_good), "normal", "bad" (src/brick-mpi.cpp:25-136) or an array of BitSet.set values"""
        if isinstance(order, str):
            order = self.skin3d(order)
        order = np.ascontiguousarray(order, dtype=np.uint64)
        return _Decomp(self.L, "orc", self.L.orc_decomp_build_order(_u3(dims_cells), ghost_cells, order, order.size))

    def skin3d(self, which="good"):
        out = np.zeros(64, dtype=np.uint64)
        n = self.L.orc_skin3d({"good": 0, "normal": 1, "bad": 2}[which], out)
        return out[:n].copy()

    def populate_rank_map(self, cart_dims, coo):
        out = np.empty(27, dtype=np.int32)
        self.L.orc_populate_rank_map((C.c_int * 3)(*cart_dims), (C.c_int * 3)(*coo), out)
        return out

    def exchange(self, dec, storages, rank_maps, step):
        n = len(storages)
        ptrs = (C.c_void_p * n)(*[s.ctypes.data for s in storages])
        rm = np.ascontiguousarray(np.asarray(rank_maps, dtype=np.int32).reshape(n, 27))
        self.L.orc_exchange(dec.h, ptrs, n, rm.reshape(-1), step)

    # ---- zmort / strong
    def zmort_get(self, zid, d, dim=3):
        return int(self.L.orc_zmort_get(zid, dim, d))

    def zmort_set(self, zid, d, p, dim=3):
        return int(self.L.orc_zmort_set(zid, dim, d, p))

    def strong_partition(self, allsubs, size, rank):
        out = np.empty(5, dtype=np.uint64)
        self.L.orc_strong_partition(allsubs, size, rank, out)
        return [int(x) for x in out]

    def strong_getrank(self, n, zid, subdim, size):
        allsubs = subdim ** 3
        _, _, split, slen, shift = self.strong_partition(allsubs, size, 0)
        dst, sub = C.c_int(), C.c_int()
        z = self.L.orc_strong_getrank(n, zid, subdim, split, slen, shift, C.byref(dst), C.byref(sub))
        return int(z), dst.value, sub.value


class _Decomp:
    """Read-only view of a BrickDecomp built by either library (prefix 'orc' or 'ref')."""

    def __init__(self, L, prefix, h):
        self.L, self.p, self.h = L, prefix, C.c_void_p(h)
        f = lambda name: getattr(L, f"{prefix}_decomp_{name}")
        info = np.empty(10, dtype=np.uint32)
        f("info")(self.h, info)
        self.sep_pos = [int(x) for x in info[0:3]]
        self.nbricks, self.grid_size, self.nruns, self.nskin = (int(x) for x in info[3:7])
        self.t_dims = [int(x) for x in info[7:10]]
        self.grid = np.empty(self.grid_size, dtype=np.uint32)
        f("grid")(self.h, self.grid)
        self.adj = np.empty((self.nbricks, 27), dtype=np.uint32)
        f("adj")(self.h, self.adj)
        self.skin_size = np.empty(self.nskin, dtype=np.int64)
        f("skin_size")(self.h, self.skin_size)
        self.ghost = np.empty((self.nruns, 7), dtype=np.uint64)
        self.skin = np.empty((self.nruns, 7), dtype=np.uint64)
        f("runs")(self.h, 0, self.ghost)
        f("runs")(self.h, 1, self.skin)

    def __del__(self):
        try:
            getattr(self.L, f"{self.p}_decomp_free")(self.h)
        except Exception:
            pass


class Ref:
    """The reference's own code (oracle/_ref/libbrickref.so)."""

    def __init__(self, path=None):
        path = path or os.path.join(HERE, "_ref", "libbrickref.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/_ref/libbrickref.so missing: run `make -C oracle ref` where /root/reference exists")
        L = self.L = C.CDLL(path)
        L.ref_init_grid.argtypes = [C.c_long * 3, _u32p, _u32p]
        for f in (L.ref_copy_to_brick, L.ref_copy_from_brick, L.ref_compare_brick):
            f.argtypes = [C.c_long * 3, C.c_long * 3, C.c_long * 3, _f64p, _u32p, _u32p, C.c_uint32, _f64p, C.c_size_t,
                          C.c_uint]
        L.ref_compare_brick.restype = C.c_int
        L.ref_brick_at.restype = C.c_double
        L.ref_brick_at.argtypes = [_u32p, C.c_uint32, _f64p, C.c_size_t, C.c_uint, C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.ref_brick_taps.argtypes = [_u32p, C.c_uint32, _f64p, _f64p, C.c_size_t, _u32p, C.c_long, _i32p, C.c_int, _f64p]
        if hasattr(L, 'ref_brick_cond'):
            L.ref_brick_cond.argtypes = [_u32p, C.c_uint32, _f64p, _f64p, C.c_size_t, _u32p, C.c_long, _f64p]
        L.ref_bitset_of.restype = C.c_uint64
        L.ref_bitset_of.argtypes = [_i32p, C.c_int]
        L.ref_bitset_neg.restype = C.c_uint64
        L.ref_bitset_neg.argtypes = [C.c_uint64]
        L.ref_allneighbors3.argtypes = [_u64p]
        L.ref_skin3d_good.argtypes = [_u64p]
        L.ref_decomp_build.restype = C.c_void_p
        L.ref_decomp_build.argtypes = [C.c_uint32 * 3, C.c_uint32]
        if hasattr(L, "ref_decomp_build_order"):
            L.ref_decomp_build_order.restype = C.c_void_p
            L.ref_decomp_build_order.argtypes = [C.c_uint32 * 3, C.c_uint32, C.c_int]
            L.ref_skin3d.argtypes = [C.c_int, _u64p]
        L.ref_decomp_free.argtypes = [C.c_void_p]
        L.ref_decomp_info.argtypes = [C.c_void_p, _u32p]
        L.ref_decomp_grid.argtypes = [C.c_void_p, _u32p]
        L.ref_decomp_adj.argtypes = [C.c_void_p, _u32p]
        L.ref_decomp_skin_size.argtypes = [C.c_void_p, _i64p]
        L.ref_decomp_runs.argtypes = [C.c_void_p, C.c_int, _u64p]
        L.ref_zmort_get.restype = C.c_uint64
        L.ref_zmort_get.argtypes = [C.c_uint64] * 3
        L.ref_zmort_set.restype = C.c_uint64
        L.ref_zmort_set.argtypes = [C.c_uint64] * 4
        for nm in ("mpi7", "mpi13", "mpi25", "mpi125"):
            getattr(L, f"ref_array_{nm}").argtypes = [_f64p, _f64p, C.c_long * 3, C.c_long * 3, C.c_long * 3]

    def init_grid(self, dims):
        size = int(dims[0]) * int(dims[1]) * int(dims[2])
        grid = np.empty(size, dtype=np.uint32)
        adj = np.empty((size, 27), dtype=np.uint32)
        self.L.ref_init_grid(_l3(dims), grid, adj)
        return grid, adj

    def copy_to_brick(self, dimlist, padding, ghost, arr, grid, adj, bdat, step, boff=0):
        self.L.ref_copy_to_brick(_l3(dimlist), _l3(padding), _l3(ghost), arr.reshape(-1), grid, adj.reshape(-1),
                                 adj.shape[0], bdat, step, boff)

    def copy_from_brick(self, dimlist, padding, ghost, arr, grid, adj, bdat, step, boff=0):
        self.L.ref_copy_from_brick(_l3(dimlist), _l3(padding), _l3(ghost), arr.reshape(-1), grid, adj.reshape(-1),
                                   adj.shape[0], bdat, step, boff)

    def compare_brick(self, dimlist, padding, ghost, arr, grid, adj, bdat, step, boff=0):
        return bool(self.L.ref_compare_brick(_l3(dimlist), _l3(padding), _l3(ghost), arr.reshape(-1), grid,
                                             adj.reshape(-1), adj.shape[0], bdat, step, boff))

    def brick_at(self, adj, bdat, step, b, k, j, i, boff=0):
        return self.L.ref_brick_at(adj.reshape(-1), adj.shape[0], bdat, step, boff, b, k, j, i)

    def brick_taps(self, adj, bin_, bout, step, ids, taps, coeff):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        taps = np.ascontiguousarray(taps, dtype=np.int32)
        coeff = np.ascontiguousarray(coeff, dtype=np.float64)
        self.L.ref_brick_taps(adj.reshape(-1), adj.shape[0], bin_, bout, step, ids, ids.size, taps.reshape(-1),
                              taps.shape[0], coeff)

    def brick_cond(self, adj, bin_, bout, step, ids, coeff):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        coeff = np.ascontiguousarray(coeff, dtype=np.float64)
        self.L.ref_brick_cond(adj.reshape(-1), adj.shape[0], bin_, bout, step, ids, ids.size, coeff)

    def array_stencil(self, name, arr_in, arr_out, lo, hi):
        ext = (arr_in.shape[2], arr_in.shape[1], arr_in.shape[0])
        getattr(self.L, f"ref_array_{name}")(arr_in.reshape(-1), arr_out.reshape(-1), _l3(ext), _l3(lo), _l3(hi))

    def st_iter(self, name):
        return getattr(self.L, f"ref_iter_{name}")()

    def bitset_of(self, elems):
        e = np.asarray(elems, dtype=np.int32)
        return int(self.L.ref_bitset_of(e, e.size))

    def bitset_neg(self, s):
        return int(self.L.ref_bitset_neg(s))

    def allneighbors3(self):
        out = np.empty(27, dtype=np.uint64)
        self.L.ref_allneighbors3(out)
        return out

    def skin3d_good(self):
        out = np.empty(32, dtype=np.uint64)
        n = self.L.ref_skin3d_good(out)
        return out[:n].copy()

    def decomp(self, dims_cells, ghost_cells=8, order="good"):
        which = {"good": 0, "normal": 1, "bad": 2}[order]
        if which == 0:
            return _Decomp(self.L, "ref", self.L.ref_decomp_build(_u3(dims_cells), ghost_cells))
        return _Decomp(self.L, "ref", self.L.ref_decomp_build_order(_u3(dims_cells), ghost_cells, which))

    def skin3d(self, which="good"):
        out = np.zeros(64, dtype=np.uint64)
        n = self.L.ref_skin3d({"good": 0, "normal": 1, "bad": 2}[which], out)
        return out[:n].copy()

    def zmort_get(self, zid, d, dim=3):
        return int(self.L.ref_zmort_get(zid, dim, d))

    def zmort_set(self, zid, d, p, dim=3):
        return int(self.L.ref_zmort_set(zid, dim, d, p))


# tap tables {dk,dj,di,coeff idx} in the textual order of the reference sums
TAPS_7PT = [(1, 0, 0, 5), (-1, 0, 0, 6), (0, 1, 0, 3), (0, -1, 0, 4), (0, 0, 1, 1), (0, 0, -1, 2), (0, 0, 0, 0)]


def taps_27pt():
    """stencils/3axis.cpp:214-233: lexicographic (k,j,i) taps with coeff[16] on (k+1,j+1,i)."""
    t = []
    c = 0
    for dk in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for di in (-1, 0, 1):
                if c == 16:
                    t.append((1, 1, 0, 16))
                else:
                    t.append((dk, dj, di, c))
                c += 1
    return t
