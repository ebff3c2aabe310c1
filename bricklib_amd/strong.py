"""Strong-scaling driver: global d^3 domain cut into (d/s)^3 subdomains in Z-Morton order, contiguous id ranges per
rank (strong/main.cpp, strong/main.cu, strong/args.cpp, include/zmort.h of the reference).

MI355X-native layout: all subdomains of a rank live in ONE brick storage with a block-diagonal adjacency table
(subdomain m's brick b is global brick m*nb+b), so a sweep over every subdomain is a single kernel launch over one
concatenated brick list -- the reference's `Brick3D barr[2*nsub]` table + 3-D grid (strong/main.cu:85-99,464-476)
without a per-subdomain indirection.  The exchange follows strong/main.cu:188-395 at run granularity:
  * ghost run of a subdomain whose source subdomain lives on this rank  -> one device link (RegionLink, :201-208)
  * otherwise -> packed into one buffer per peer rank (pack links), ONE send + ONE recv per peer, unpack links.
Both sides enumerate a peer's runs in the same canonical order (receiver Z-Morton id, run index), so the buffers
need no header.  All link tables go through brk_link_copy.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib
from .brick import Brick, BrickInfo, BrickStorage, grid_to_device
from .decomp import SP_RECV, SP_SEND, SP_STORAGE, BrickDecomp, Exchanger, XPlan, exchange_loopback
from .order import decomp_lists
from .stencils import apply_stencil, stencil_iters


def zmort_coords(zid):
    return [int(lib.brk_zmort_get(zid, a)) for a in range(3)]


def partition(allsubs, size, rank):
    lr = (C.c_uint64 * 2)()
    lib.brk_strong_partition(allsubs, size, rank, lr)
    return int(lr[0]), int(lr[1])


def neighbor_of(nset, zid, subdim, allsubs, size):
    """(neighbour zmort id, owner rank, slot on owner) -- getrank() of strong/args.cpp:36-56"""
    o, s = C.c_int(), C.c_int()
    nz = lib.brk_strong_neighbor(nset, zid, subdim, allsubs, size, C.byref(o), C.byref(s))
    return int(nz), o.value, s.value


class StrongRank:
    def __init__(self, dom, sdom, rank, size, device, rows_per_block=8):
        assert dom % sdom == 0 and (dom // sdom) & (dom // sdom - 1) == 0, "blocks per axis must be a power of two"
        self.dom, self.sdom, self.rank, self.size, self.device = dom, sdom, rank, size, torch.device(device)
        self.subdim = dom // sdom
        self.allsubs = self.subdim ** 3
        self.l, self.r = partition(self.allsubs, size, rank)
        self.nsub = self.r - self.l
        d = self.d = BrickDecomp((sdom, sdom, sdom), 8)        # ONE decomposition shared by all subdomains (strong/main.cpp:91-94)
        nb = self.nb = d.nbricks
        # block-diagonal adjacency: brick ids of subdomain m are shifted by m*nb
        adj = np.concatenate([d.adj.astype(np.uint64) + m * nb for m in range(self.nsub)]).astype(np.uint32)
        self.info = BrickInfo(adj, self.device)
        self.storages = [BrickStorage.allocate(self.nsub * nb, 512, self.device) for _ in range(3)]
        self.bricks = [Brick(self.info, st, 0) for st in self.storages]
        lists = decomp_lists(d, rows_per_block)
        shift = lambda ids: np.concatenate([ids.astype(np.uint64) + m * nb for m in range(self.nsub)]).astype(np.uint32)
        self.list_all = grid_to_device(shift(lists["all"]), self.device)
        self.list_domain = grid_to_device(shift(lists["domain"]), self.device)
        self.grid_dev = grid_to_device(d.grid.reshape(-1), self.device)
        self._plan()

    # ---- exchange plan (strong/main.cu:188-395, run granularity) ----------------------------------------------
    def _plan(self):
        """Tables for brk_xplan_build_tables: same-rank ghost runs are storage->storage links, the others are packed per
        peer rank (canonical order on both sides: receiver Z-Morton id, run index), ONE send + ONE recv per peer."""
        d, nb, esz = self.d, self.nb, 8
        off = lambda m, pos: (m * nb + pos) * 512 * esz          # byte offset inside storages[0]
        copies = []
        recv, send = {}, {}       # peer -> list of (key=(receiver zmort, run), sub index, run)
        for m in range(self.nsub):
            z = self.l + m
            for q, (g, s) in enumerate(zip(d.ghost, d.skin)):
                # receive side: ghost run q of z comes from the subdomain in direction g.neighbor
                zs, owner, slot = neighbor_of(g["neighbor"], z, self.subdim, self.allsubs, self.size)
                if owner == self.rank:
                    copies.append(dict(stage=0, phase=0, src_space=SP_STORAGE, dst_space=SP_STORAGE, src_peer=-1, dst_peer=-1,
                                       src_off=off(slot, s["pos"]), dst_off=off(m, g["pos"]), bytes=g["len"] * 512 * esz))
                else:
                    recv.setdefault(owner, []).append(((z, q), m, q))
                # send side: skin run q of z goes to the subdomain in direction s.neighbor
                zr, owner_r, _ = neighbor_of(s["neighbor"], z, self.subdim, self.allsubs, self.size)
                if owner_r != self.rank:
                    send.setdefault(owner_r, []).append(((zr, q), m, q))
        self.peers = sorted(set(recv) | set(send))
        msgs = []
        for p in self.peers:
            so = ro = 0
            for _, m, q in sorted(send.get(p, [])):
                n = d.skin[q]["len"] * 512 * esz
                copies.append(dict(stage=0, phase=0, src_space=SP_STORAGE, dst_space=SP_SEND, src_peer=-1, dst_peer=p,
                                   src_off=off(m, d.skin[q]["pos"]), dst_off=so, bytes=n))
                so += n
            for _, m, q in sorted(recv.get(p, [])):
                n = d.ghost[q]["len"] * 512 * esz
                copies.append(dict(stage=0, phase=1, src_space=SP_RECV, dst_space=SP_STORAGE, src_peer=p, dst_peer=-1,
                                   src_off=ro, dst_off=off(m, d.ghost[q]["pos"]), bytes=n))
                ro += n
            if ro:
                msgs.append(dict(stage=0, is_send=0, peer=p, space=SP_RECV, off=0, bytes=ro))
            if so:
                msgs.append(dict(stage=0, is_send=1, peer=p, space=SP_SEND, off=0, bytes=so))
        plan = XPlan.from_tables(self.rank, copies, msgs)
        # the view: buffers + device link tables inside libbrk.so; begin / transport / end are C-ABI calls
        self.xch = Exchanger(d, self.storages[0], self.rank, mode="packed", plan=plan)
        self.send_buf, self.recv_buf = self.xch.send_buf, self.xch.recv_buf
        self.bytes_remote = plan.bytes_remote

    # ---- data movement ----------------------------------------------------------------------------------------
    def origin(self, m):
        """global cell coordinate (i,j,k) of subdomain slot m ('+d' = coordinate+1, strong/args.cpp:36-48)"""
        return [c * self.sdom for c in zmort_coords(self.l + m)]

    def sub_brick(self, which, m):
        """Brick view of one subdomain inside storage `which` (for copyToBrick/copyFromBrick with d.grid)."""
        st = self.storages[which]
        view = BrickStorage(st.dat[m * self.nb * 512:(m + 1) * self.nb * 512], self.nb, 512)
        return Brick(self.d.getBrickInfo(self.device), view, 0)

    def exchange_begin(self, stream=None):
        """same-rank links (strong/main.cu:440-443) and pack (:479-482): one launch"""
        self.xch.exchange_begin(stream)

    def exchange_end(self, stream=None):
        """unpack (strong/main.cu:459-462)"""
        self.xch.exchange_end(stream)

    def exchange(self, transport, stream=None):
        """transport(self, stream) moves send_buf[p] of this rank into recv_buf[self.rank] of peer p."""
        self.exchange_begin(stream)
        if self.peers:
            transport(self, stream)
        self.exchange_end(stream)

    def sweeps(self, stencil, coeff=None, stream=None):
        """strong/main.cpp:359-365: 1<-0; (2<-1; 1<-2) x (S/2-1); 0<-1 without the outer brick ring."""
        S = stencil_iters(stencil)
        b = self.bricks
        apply_stencil(stencil, b[0], b[1], ids=self.list_all, coeff=coeff, stream=stream)
        for _ in range(S // 2 - 1):
            apply_stencil(stencil, b[1], b[2], ids=self.list_all, coeff=coeff, stream=stream)
            apply_stencil(stencil, b[2], b[1], ids=self.list_all, coeff=coeff, stream=stream)
        apply_stencil(stencil, b[1], b[0], ids=self.list_domain, coeff=coeff, stream=stream)
        return S


def rccl_transport(comm):
    """the product transport: ncclGroupStart; ncclRecv/ncclSend per peer; ncclGroupEnd inside libbrk.so (brk_exchange_transport)
    on a communicator created through the C ABI (bricklib_amd.decomp.Comm)"""
    def go(sr, stream=None):
        sr.xch.comm = comm
        sr.xch.exchange_transport(stream)
    return go


def nccl_transport(sr, stream=None):
    """the same messages over torch.distributed (backend nccl = RCCL): batch_isend_irecv on tensors aliasing the view's buffers"""
    sr.xch.comm, sr.xch.staged = None, False
    sr.xch.exchange_transport(stream)


def loopback_transport(all_ranks):
    """Transport for several virtual ranks living on one GPU (brk_exchange_loopback): every send is copied into the matching
    receive of the destination rank's view.  Runs once all ranks have packed: call with the last rank."""
    last = [r for r in all_ranks if r.peers][-1:]

    def go(sr, stream=None):
        if last and sr is last[0]:
            exchange_loopback([r.xch for r in all_ranks], 0, stream)
    return go


def staged_transport(sr, stream=None):
    """REHEARSAL ONLY (bench.py --backend gloo): ranks share one GPU, buffers detour through host memory and gloo."""
    sr.xch.comm, sr.xch.staged = None, True
    sr.xch.exchange_transport(stream)
