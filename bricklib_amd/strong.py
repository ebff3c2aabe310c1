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
from .decomp import BrickDecomp, LinkTable
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
        d, nb, esz = self.d, self.nb, 8
        st0 = self.storages[0]
        base = st0.dat.data_ptr()
        off = lambda m, pos: base + (m * nb + pos) * 512 * esz
        local = []
        recv, send = {}, {}       # peer -> list of (key=(receiver zmort, run), sub index, run)
        for m in range(self.nsub):
            z = self.l + m
            for q, (g, s) in enumerate(zip(d.ghost, d.skin)):
                # receive side: ghost run q of z comes from the subdomain in direction g.neighbor
                zs, owner, slot = neighbor_of(g["neighbor"], z, self.subdim, self.allsubs, self.size)
                if owner == self.rank:
                    local.append((off(slot, s["pos"]), off(m, g["pos"]), g["len"] * 512 * esz))
                else:
                    recv.setdefault(owner, []).append(((z, q), m, q))
                # send side: skin run q of z goes to the subdomain in direction s.neighbor
                zr, owner_r, _ = neighbor_of(s["neighbor"], z, self.subdim, self.allsubs, self.size)
                if owner_r != self.rank:
                    send.setdefault(owner_r, []).append(((zr, q), m, q))
        self.local_table = LinkTable(local, self.device)
        self.peers = sorted(set(recv) | set(send))
        self.send_buf, self.recv_buf = {}, {}
        pack, unpack = [], []
        for p in self.peers:
            rl, sl = sorted(recv.get(p, [])), sorted(send.get(p, []))
            rbytes = sum(d.ghost[q]["len"] for _, _, q in rl) * 512 * esz
            sbytes = sum(d.skin[q]["len"] for _, _, q in sl) * 512 * esz
            self.recv_buf[p] = torch.empty(rbytes // esz, dtype=torch.float64, device=self.device)
            self.send_buf[p] = torch.empty(sbytes // esz, dtype=torch.float64, device=self.device)
            o = self.send_buf[p].data_ptr()
            for _, m, q in sl:
                n = d.skin[q]["len"] * 512 * esz
                pack.append((off(m, d.skin[q]["pos"]), o, n))
                o += n
            o = self.recv_buf[p].data_ptr()
            for _, m, q in rl:
                n = d.ghost[q]["len"] * 512 * esz
                unpack.append((o, off(m, d.ghost[q]["pos"]), n))
                o += n
        self.pack_table = LinkTable(pack, self.device)
        self.unpack_table = LinkTable(unpack, self.device)
        self.bytes_remote = sum(b.numel() for b in self.send_buf.values()) * esz

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
        """same-rank links (strong/main.cu:440-443) and pack (:479-482)"""
        self.local_table.run(stream)
        self.pack_table.run(stream)

    def exchange_end(self, stream=None):
        """unpack (strong/main.cu:459-462)"""
        self.unpack_table.run(stream)

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


def nccl_transport(sr, stream=None):
    """one send + one recv per peer rank over torch.distributed (RCCL)"""
    import torch.distributed as dist
    ops = []
    for p in sr.peers:
        if sr.recv_buf[p].numel():
            ops.append(dist.P2POp(dist.irecv, sr.recv_buf[p], p))
        if sr.send_buf[p].numel():
            ops.append(dist.P2POp(dist.isend, sr.send_buf[p], p))
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for r in dist.batch_isend_irecv(ops):
            r.wait()


def loopback_transport(all_ranks):
    """Transport for several virtual ranks living on one GPU: peer p's recv buffer for r is r's send buffer for p."""
    def go(sr, stream=None):
        for p in sr.peers:
            all_ranks[p].recv_buf[sr.rank].copy_(sr.send_buf[p])
    return go


def staged_transport(sr, stream=None):
    """REHEARSAL ONLY (bench.py --backend gloo): ranks share one GPU, buffers detour through host memory and gloo."""
    import torch.distributed as dist
    (stream or torch.cuda.current_stream()).synchronize()
    host_s = {p: sr.send_buf[p].cpu() for p in sr.peers}
    host_r = {p: torch.empty(sr.recv_buf[p].numel(), dtype=torch.float64) for p in sr.peers}
    ops = []
    for p in sr.peers:
        if host_r[p].numel():
            ops.append(dist.P2POp(dist.irecv, host_r[p], p))
        if host_s[p].numel():
            ops.append(dist.P2POp(dist.isend, host_s[p], p))
    for r in dist.batch_isend_irecv(ops):
        r.wait()
    with torch.cuda.stream(stream or torch.cuda.current_stream()):
        for p in sr.peers:
            sr.recv_buf[p].copy_(host_r[p])
