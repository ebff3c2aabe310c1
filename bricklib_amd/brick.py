"""Brick data structures, mirroring include/brick.h, include/bricksetup.h, include/brickcompare.h and
include/brick-gpu.h of the reference (names and argument meaning kept)."""
import ctypes as C

import numpy as np
import torch

import os
import sys

from ._lib import PlacePartner, Placement, check, l3, lib

BRICK_TOLERANCE = 1e-6  # include/cmpconst.h:9
BDIM = (8, 8, 8)
BSIZE = 512


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(stream=None):
    if stream is None:
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


def _as_u32_tensor(a, device):
    """uint32 host array -> int32 tensor with the same bits (ids < 2^31)."""
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return torch.from_numpy(a.view(np.int32).copy()).to(device)


class BrickInfo:
    """Adjacency list of a set of bricks (include/brick.h:96-127).  adj[b][(dk+1)*9+(dj+1)*3+(di+1)]."""

    def __init__(self, adj, device=None):
        if isinstance(adj, np.ndarray):
            self.adj_host = np.ascontiguousarray(adj.reshape(-1, 27), dtype=np.uint32)
            self.adj = None
        else:
            self.adj_host = None
            self.adj = adj
        self.nbricks = (self.adj_host if self.adj_host is not None else self.adj).shape[0]
        if device is not None:
            self.to(device)

    def to(self, device):
        """The device copy of a table that came from the host (init_grid, BrickDecomp.getBrickInfo, movBrickInfo) belongs to this
        object and is never written again -- the reference's BrickInfo is filled once by its builders -- so it is registered
        with the library as immutable (brk_adjacency_register): kernels may then skip adjacency rows they can predict.  A table
        handed in as a device tensor stays unregistered (its owner may rewrite it)."""
        if self.adj is None or self.adj.device != torch.device(device):
            self._release()
            self.adj = _as_u32_tensor(self.adj_host, device).reshape(-1, 27)
            if self.adj.device.type == "cuda":
                import weakref
                ptr, dev = self.adj.data_ptr(), self.adj.device
                with torch.cuda.device(dev):
                    check(lib.brk_adjacency_register(C.c_void_p(ptr), self.nbricks))
                keep = self.adj               # the finalizer holds the tensor: the table outlives its registration

                def fin(ptr=ptr, dev=dev, keep=keep):
                    try:
                        with torch.cuda.device(dev):
                            lib.brk_adjacency_release(C.c_void_p(ptr))
                    except Exception:
                        pass
                self._fin = weakref.finalize(self, fin)
        return self

    def _release(self):
        fin = getattr(self, "_fin", None)
        if fin is not None:
            fin()
            self._fin = None

    def allocate(self, step, device="cuda"):
        return BrickStorage.allocate(self.nbricks, step, device)


class _PlacedMemory:
    """Device memory from brk_storage_alloc, handed to torch through __cuda_array_interface__ (torch keeps this object alive for
    as long as a tensor aliases the memory) and given back with brk_storage_free when the last tensor is gone."""

    def __init__(self, nelem, device, partners=None):
        self.device, self.nelem = device, int(nelem)
        ptr, rep = C.c_void_p(), Placement()
        with torch.cuda.device(device):
            if partners is None:
                check(lib.brk_storage_alloc(self.nelem * 8, C.byref(ptr), C.byref(rep)))
            else:   # [(placed BrickStorage, "opposite" | "same")]: brk_storage_alloc_ex, probes on torch's current stream
                tab = (PlacePartner * max(1, len(partners)))()
                for q, (st, rel) in enumerate(partners):
                    if st.placement is None:
                        raise ValueError("BrickStorage.allocate: a partner must be a placed storage (BrickStorage.allocate of at least PLACE_MIN_BYTES)")
                    tab[q].storage_dev, tab[q].relation = st.dat.data_ptr(), {"opposite": 0, "same": 1}[rel]
                check(lib.brk_storage_alloc_ex(self.nelem * 8, tab, len(partners), C.c_void_p(torch.cuda.current_stream(device).cuda_stream),
                                               C.byref(ptr), C.byref(rep)))
        self.ptr = ptr.value
        self.report = {"candidates": rep.candidates, "chosen": rep.chosen, "probe_bricks": rep.probe_bricks, "spacer_gib": rep.spacer_gib,
                       "probe_sweep_ms_chosen": rep.ms_chosen, "probe_sweep_ms_worst": rep.ms_worst,
                       "probe_sweep_ms_same_allocation": rep.ms_same_allocation}
        if rep.candidates > 0 and rep.ms_same_allocation > 0:
            # did the search see the other kind?  (a chosen pair within 1 % of a pair inside ONE allocation: it did not -- the storage is
            # then as slow as an unplaced one, and the line that reports it says so)
            self.report["both_kinds_seen"] = bool(rep.ms_chosen < 0.99 * rep.ms_same_allocation)
        self.__cuda_array_interface__ = {"shape": (self.nelem,), "typestr": "<f8", "data": (self.ptr, False), "version": 2}

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr and not sys.is_finalizing():      # at interpreter exit the process teardown returns the memory; no HIP calls from there
            try:
                with torch.cuda.device(self.device):
                    lib.brk_storage_free(C.c_void_p(ptr))
            except Exception:
                pass


def place_min_bytes():
    """brk_storage_alloc measures storages from this size on: 512 MiB -- a probe over less runs out of the 256-MiB Infinity Cache
    (include/brk.h; BRK_PLACEMENT_MIN_BYTES overrides it, as in the library: tests of the machinery on small storages)"""
    v = os.environ.get("BRK_PLACEMENT_MIN_BYTES")
    return int(v) if v and int(v) > 0 else 512 << 20


class BrickStorage:
    """Flat storage of `chunks` chunks of `step` elements (include/brick.h:53-82).

    Device storages of 512 MiB and more are PLACED (brk_storage_alloc, include/brk.h): the device allocations of a process fall into
    two kinds, and a sweep between two storages of the same kind runs 2.7 % slower than between storages of different kinds, so the
    allocation is a measured one -- see bricklib_amd/csrc/brk_place.hip.  `placement` holds what was measured (None for small,
    host or BRK_PLACEMENT=0 storages, which come from torch's allocator as before).  By default a storage is placed against the one
    allocated before it as a read / write pair; `partners=[(storage, "opposite" | "same"), ...]` names the storages a sweep will pair
    it with and how ("same": two fields a sweep reads together)."""

    def __init__(self, dat, chunks, step, placement=None):
        self.dat, self.chunks, self.step, self.placement = dat, int(chunks), int(step), placement

    @staticmethod
    def allocate(chunks, step, device="cuda", partners=None):
        n = int(chunks) * int(step)
        dev = torch.device(device)
        if dev.type == "cuda" and n * 8 >= place_min_bytes() and os.environ.get("BRK_PLACEMENT", "1") != "0":
            if dev.index is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            mem = _PlacedMemory(n, dev, partners)
            dat = torch.as_tensor(mem, device=dev)
            if dat.data_ptr() != mem.ptr or dat.dtype != torch.float64 or dat.numel() != n:
                raise RuntimeError("BrickStorage.allocate: torch did not alias the placed storage")
            dat.zero_()
            return BrickStorage(dat, chunks, step, mem.report)
        return BrickStorage(torch.zeros(n, dtype=torch.float64, device=device), chunks, step)


class Brick:
    """One field inside a BrickStorage (include/brick.h:353-395): brick b starts at dat[offset + b*step]."""

    def __init__(self, bInfo, bStorage, offset=0):
        self.bInfo, self.bStorage, self.offset = bInfo, bStorage, int(offset)
        self.step = bStorage.step
        self.dat = bStorage.dat[self.offset:]

    def view(self):
        """[nbricks, 8, 8, 8] strided view (k, j, i)."""
        return torch.as_strided(self.dat, (self.bStorage.chunks, 8, 8, 8), (self.step, 64, 8, 1))


def init_grid(dimlist, device=None):
    """include/bricksetup.h:73-90 -> (grid uint32[size] host array, BrickInfo)."""
    size = int(dimlist[0]) * int(dimlist[1]) * int(dimlist[2])
    grid = np.empty(size, dtype=np.uint32)
    adj = np.empty((size, 27), dtype=np.uint32)
    check(lib.brk_init_grid(l3(dimlist), grid.ctypes.data_as(C.POINTER(C.c_uint32)),
                            adj.ctypes.data_as(C.POINTER(C.c_uint32))))
    return grid, BrickInfo(adj, device)


def movBrickInfo(bInfo, device="cuda"):
    """include/brick-gpu.h:38-51"""
    return BrickInfo(bInfo.adj_host, device)


def movBrickStorage(bStorage, device="cuda"):
    """include/brick-gpu.h:80-99"""
    return BrickStorage(bStorage.dat.to(device), bStorage.chunks, bStorage.step)


def grid_to_device(grid, device="cuda"):
    return _as_u32_tensor(grid, device)


def copyToBrick(dimlist, padding, ghost, arr, grid_dev, brick, stream=None):
    """include/bricksetup.h:172-200 on device: arr (array layout, i contiguous) -> bricks."""
    check(lib.brk_array_to_brick(l3(dimlist), l3(padding), l3(ghost), _ptr(arr), _ptr(grid_dev), _ptr(brick.dat),
                                 brick.step, _stream(stream)))


def copyFromBrick(dimlist, padding, ghost, arr, grid_dev, brick, stream=None):
    """include/bricksetup.h:213-221 on device."""
    check(lib.brk_brick_to_array(l3(dimlist), l3(padding), l3(ghost), _ptr(arr), _ptr(grid_dev), _ptr(brick.dat),
                                 brick.step, _stream(stream)))


def compareBrick(dimlist, padding, ghost, arr, grid_dev, brick, tol=BRICK_TOLERANCE, stream=None, return_max=False):
    """include/brickcompare.h:30-57 on device -> True when every element matches under the reference rule."""
    res = torch.zeros(2, dtype=torch.int64, device=brick.dat.device)
    check(lib.brk_compare_brick(l3(dimlist), l3(padding), l3(ghost), _ptr(arr), _ptr(grid_dev), _ptr(brick.dat),
                                brick.step, tol, _ptr(res), _stream(stream)))
    r = res.cpu().numpy()
    ok = int(r[0]) == 0
    if return_max:
        return ok, float(r[1:2].view(np.float64)[0])
    return ok
