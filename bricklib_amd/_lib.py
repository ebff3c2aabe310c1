"""ctypes binding of libbrk.so (C ABI: include/brk.h).  There is no fallback: if the HIP library has not
been built (python -c 'import __graft_entry__ as g; g.build()'  or  make -C bricklib_amd/csrc) importing the
package raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
BRK_SWEEP_KPAIRED, BRK_SWEEP_REVERSE = 1, 2   # include/brk.h
LIB_PATH = os.path.join(HERE, "csrc", "libbrk.so")

BRK_7PT_COEFF, BRK_27PT_COEFF, BRK_MPI7, BRK_MPI13, BRK_MPI25, BRK_MPI125 = range(6)


class BrkError(RuntimeError):
    pass


class Region(C.Structure):
    """brk_region == g_region of include/brick-mpi.h:185-193"""
    _fields_ = [("neighbor", C.c_uint64), ("skin_st", C.c_uint32), ("skin_ed", C.c_uint32), ("pos", C.c_uint32),
                ("len", C.c_uint32), ("first_pad", C.c_uint32), ("last_pad", C.c_uint32)]


class PlacePartner(C.Structure):
    """brk_place_partner: a live placed storage and the wanted relation (0 = BRK_PLACE_OPPOSITE, 1 = BRK_PLACE_SAME)"""
    _fields_ = [("storage_dev", C.c_void_p), ("relation", C.c_int)]


class ExprOp(C.Structure):
    """brk_expr_op: one operation of a stencil given as a register program (brk_stencil_apply_expr)"""
    _fields_ = [("code", C.c_uint32), ("dst", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("c", C.c_uint32),
                ("offset", C.c_int32 * 3), ("imm", C.c_double)]


class Placement(C.Structure):
    """brk_placement: what brk_storage_alloc measured while it placed a storage"""
    _fields_ = [("candidates", C.c_int), ("chosen", C.c_int), ("probe_bricks", C.c_int), ("spacer_gib", C.c_int), ("ms_chosen", C.c_double), ("ms_worst", C.c_double), ("ms_same_allocation", C.c_double)]


class Link(C.Structure):
    """brk_link"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("bytes", C.c_uint64)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise BrkError(f"{LIB_PATH} is missing: the HIP extension must be built (make -C bricklib_amd/csrc); "
                       "bricklib_amd has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, u32p, f64p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double)
    L3 = C.c_long * 3
    L.brk_last_error.restype = C.c_char_p
    L.brk_abi_version.restype = C.c_int
    for f in (L.brk_stencil_radius, L.brk_stencil_ncoeff, L.brk_stencil_iters):
        f.argtypes = [C.c_int]
        f.restype = C.c_int
    L.brk_stencil_apply_range.argtypes = [C.c_int, vp, vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, f64p, vp]
    L.brk_stencil_apply_list.argtypes = [C.c_int, vp, vp, vp, C.c_size_t, vp, C.c_uint32, f64p, vp]
    L.brk_stencil_apply_list_kpaired.argtypes = [C.c_int, vp, vp, vp, C.c_size_t, vp, C.c_uint32, f64p, vp]
    U3 = C.c_uint32 * 3
    L.brk_stencil_apply_box.argtypes = [C.c_int, vp, vp, vp, C.c_size_t, C.c_uint32, U3, U3, U3, f64p, vp]
    L.brk_stencil_apply_range_ex.argtypes = [C.c_int, vp, vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, f64p, C.c_uint32, vp]
    L.brk_stencil_apply_list_ex.argtypes = [C.c_int, vp, vp, vp, C.c_size_t, vp, C.c_uint32, f64p, C.c_uint32, vp]
    L.brk_stencil_apply_box_ex.argtypes = [C.c_int, vp, vp, vp, C.c_size_t, C.c_uint32, U3, U3, U3, f64p, C.c_uint32, vp]
    L.brk_stencil_apply_taps.argtypes = [vp, vp, vp, C.c_size_t, vp, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.POINTER(C.c_int32), f64p, vp]
    if hasattr(L, "brk_set_variant"):      # development builds only (make DEV=1 / VARIANTS=1): not part of the product ABI
        L.brk_set_variant.argtypes = [C.c_int, C.c_int]
    L.brk_link_copy.argtypes = [vp, C.c_uint32, C.c_uint64, vp]
    L.brk_storage_alloc.argtypes = [C.c_size_t, C.POINTER(vp), C.POINTER(Placement)]
    L.brk_storage_alloc_ex.argtypes = [C.c_size_t, C.POINTER(PlacePartner), C.c_uint32, vp, C.POINTER(vp), C.POINTER(Placement)]
    L.brk_storage_free.argtypes = [vp]
    L.brk_placement_probe.argtypes = [vp, vp, C.c_size_t, vp, f64p]
    L.brk_array_to_brick.argtypes = [L3, L3, L3, vp, vp, vp, C.c_size_t, vp]
    L.brk_brick_to_array.argtypes = [L3, L3, L3, vp, vp, vp, C.c_size_t, vp]
    L.brk_compare_brick.argtypes = [L3, L3, L3, vp, vp, vp, C.c_size_t, C.c_double, vp, vp]
    L.brk_init_grid.argtypes = [L3, u32p, u32p]
    L.brk_decomp_build.argtypes = [C.c_uint32 * 3, C.c_uint32, C.POINTER(vp)]
    L.brk_decomp_free.argtypes = [vp]
    L.brk_decomp_free.restype = None
    for name in ("nbricks", "grid_size", "nruns", "nskin"):
        f = getattr(L, "brk_decomp_" + name)
        f.argtypes = [vp]
        f.restype = C.c_uint32
    L.brk_decomp_tdims.argtypes = [vp, C.c_uint32 * 3]
    L.brk_decomp_sep_pos.argtypes = [vp, C.c_uint32 * 3]
    L.brk_decomp_grid.argtypes = [vp]
    L.brk_decomp_grid.restype = u32p
    L.brk_decomp_adj.argtypes = [vp]
    L.brk_decomp_adj.restype = u32p
    L.brk_decomp_ghost.argtypes = [vp]
    L.brk_decomp_ghost.restype = C.POINTER(Region)
    L.brk_decomp_skin.argtypes = [vp]
    L.brk_decomp_skin.restype = C.POINTER(Region)
    L.brk_decomp_skinlist.argtypes = [vp]
    L.brk_decomp_skinlist.restype = C.POINTER(C.c_uint64)
    L.brk_decomp_skin_size.argtypes = [vp]
    L.brk_decomp_skin_size.restype = C.POINTER(C.c_int64)
    L.brk_allneighbors3.argtypes = [C.c_uint64 * 27]
    L.brk_allneighbors3.restype = None
    L.brk_populate_rank_map.argtypes = [C.c_int * 3, C.c_int * 3, C.c_int * 27]
    L.brk_dims_create.argtypes = [C.c_int, C.c_int * 3]
    L.brk_zmort_get.argtypes = [C.c_uint64, C.c_uint64]
    L.brk_zmort_get.restype = C.c_uint64
    L.brk_zmort_set.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    L.brk_zmort_set.restype = C.c_uint64
    L.brk_strong_partition.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_uint64 * 2]
    L.brk_strong_partition.restype = None
    L.brk_strong_neighbor.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]
    L.brk_strong_neighbor.restype = C.c_uint64
    # ghost-zone exchange (plans, views, communicators)
    i27 = C.c_int * 27
    L.brk_xplan_build_decomp.argtypes = [vp, i27, C.c_int, C.c_int, C.c_uint, C.c_size_t, C.POINTER(vp)]
    L.brk_xplan_build_tables.argtypes = [C.c_int, C.c_uint32, vp, C.c_uint32, vp, C.POINTER(vp)]
    L.brk_xplan_free.argtypes = [vp]
    L.brk_xplan_free.restype = None
    L.brk_xplan_my_rank.argtypes = [vp]
    for name in ("nstages", "ncopies", "nmsgs"):
        f = getattr(L, "brk_xplan_" + name)
        f.argtypes = [vp]
        f.restype = C.c_uint32
    from .decomp_types import XCopy, XMsg
    L.brk_xplan_copies.argtypes = [vp]
    L.brk_xplan_copies.restype = C.POINTER(XCopy)
    L.brk_xplan_msgs.argtypes = [vp]
    L.brk_xplan_msgs.restype = C.POINTER(XMsg)
    for name in ("bytes_remote", "bytes_local"):
        f = getattr(L, "brk_xplan_" + name)
        f.argtypes = [vp]
        f.restype = C.c_uint64
    L.brk_xplan_buffer_bytes.argtypes = [vp, C.c_uint32, C.c_int, C.c_uint32]
    L.brk_xplan_buffer_bytes.restype = C.c_uint64
    L.brk_exchange_view_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.brk_exchange_view_free.argtypes = [vp]
    L.brk_exchange_view_free.restype = None
    L.brk_exchange_begin.argtypes = [vp, C.c_uint32, vp]
    L.brk_exchange_transport.argtypes = [vp, C.c_uint32, vp, vp]
    L.brk_exchange_end.argtypes = [vp, C.c_uint32, vp]
    L.brk_exchange_run.argtypes = [vp, vp, vp]
    L.brk_exchange_loopback.argtypes = [C.POINTER(vp), C.c_uint32, C.c_uint32, vp]
    L.brk_exchange_view_buffer.argtypes = [vp, C.c_uint32, C.c_int, C.c_uint32]
    L.brk_exchange_view_buffer.restype = vp
    L.brk_exchange.argtypes = [vp, vp, C.c_size_t, vp, i27, vp]
    L.brk_comm_unique_id.argtypes = [vp]
    L.brk_comm_init_rank.argtypes = [C.POINTER(vp), C.c_int, vp, C.c_int]
    L.brk_comm_destroy.argtypes = [vp]
    L.brk_comm_rank.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.brk_decomp_build_order.argtypes = [C.c_uint32 * 3, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(vp)]
    L.brk_skin3d_order.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.brk_comm_library.restype = C.c_char_p
    L.brk_comm_library.argtypes = []
    L.brk_adjacency_register.argtypes = [vp, C.c_uint32]
    L.brk_adjacency_release.argtypes = [vp]
    L.brk_stencil_apply_expr.argtypes = [vp, vp, vp, C.c_size_t, vp, C.c_uint32, C.c_uint32, C.POINTER(ExprOp), C.c_uint32, C.c_uint32, vp]
    L.brk_expr_compile.argtypes = [C.POINTER(ExprOp), C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.brk_expr_apply.argtypes = [vp, vp, vp, vp, C.c_size_t, vp, C.c_uint32, C.c_uint32, vp]
    L.brk_expr_apply_box.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_uint32, C.c_uint32 * 3, C.c_uint32 * 3, C.c_uint32 * 3, vp]
    L.brk_expr_release.argtypes = [vp]
    L.brk_expr_source.argtypes = [C.POINTER(ExprOp), C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.brk_expr_kernel_source.argtypes = [vp]
    L.brk_expr_kernel_source.restype = C.c_char_p
    L.brk_expr_kernel_from_disk.argtypes = [vp]
    L.brk_finalize.argtypes = []
    L.brk_last_kernel.restype = C.c_char_p
    L.brk_last_kernel.argtypes = []
    return L


lib = _load()

# every symbol include/brk.h declares (tests check the library exports each of them)
ABI_SYMBOLS = [
    "brk_last_error", "brk_abi_version", "brk_stencil_radius", "brk_stencil_ncoeff", "brk_stencil_iters",
    "brk_stencil_apply_range", "brk_stencil_apply_list", "brk_stencil_apply_list_kpaired", "brk_stencil_apply_box", "brk_stencil_apply_range_ex", "brk_stencil_apply_list_ex", "brk_stencil_apply_box_ex", "brk_stencil_apply_taps", "brk_stencil_apply_expr", "brk_expr_compile", "brk_expr_apply", "brk_expr_apply_box", "brk_expr_release", "brk_expr_source",
               "brk_expr_kernel_source", "brk_expr_kernel_from_disk", "brk_link_copy", "brk_array_to_brick",
    "brk_brick_to_array", "brk_compare_brick", "brk_init_grid", "brk_decomp_build", "brk_decomp_free",
    "brk_decomp_nbricks", "brk_decomp_grid_size", "brk_decomp_tdims", "brk_decomp_sep_pos", "brk_decomp_grid",
    "brk_decomp_adj", "brk_decomp_nruns", "brk_decomp_ghost", "brk_decomp_skin", "brk_decomp_nskin",
    "brk_decomp_skinlist", "brk_decomp_skin_size", "brk_allneighbors3", "brk_populate_rank_map", "brk_dims_create",
    "brk_zmort_get", "brk_zmort_set", "brk_strong_partition", "brk_strong_neighbor",
    "brk_xplan_build_decomp", "brk_xplan_build_tables", "brk_xplan_free", "brk_xplan_my_rank", "brk_xplan_nstages",
    "brk_xplan_ncopies", "brk_xplan_copies", "brk_xplan_nmsgs", "brk_xplan_msgs", "brk_xplan_bytes_remote",
    "brk_xplan_bytes_local", "brk_xplan_buffer_bytes", "brk_exchange_view_create", "brk_exchange_view_free",
    "brk_exchange_begin", "brk_exchange_transport", "brk_exchange_end", "brk_exchange_run", "brk_exchange_loopback",
    "brk_exchange_view_buffer", "brk_exchange", "brk_comm_unique_id", "brk_comm_init_rank", "brk_comm_destroy", "brk_comm_rank", "brk_comm_library",
    "brk_adjacency_register", "brk_adjacency_release", "brk_decomp_build_order", "brk_skin3d_order",
    "brk_storage_alloc", "brk_storage_alloc_ex", "brk_storage_free", "brk_placement_probe", "brk_finalize", "brk_last_kernel",
]


def check(rc):
    if rc != 0:
        raise BrkError(f"libbrk error {rc}: {lib.brk_last_error().decode()}")


def l3(v):
    return (C.c_long * 3)(*[int(x) for x in v])
