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

alias_local (default): a ghost run whose source subdomain lives on this rank is not copied at all.  Bricks find their
neighbours through the adjacency table, so every reference to such a ghost brick is pointed at the source subdomain's own
brick instead (the adjacency-table form of the reference's memory-mapped ghost zones, weak/mmap.cu / strong/mmap.cu); the
aliased ghost bricks are then neither exchanged nor swept.  The subdomain cells come out bit-identical: what a sweep over a
ghost brick recomputes inside its valid cone is exactly what the owning subdomain computes for that brick.  With every
subdomain on one rank the cycle degenerates to ST_ITER sweeps of the periodic global domain and no exchange.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib
from .brick import Brick, BrickInfo, BrickStorage, grid_to_device
from .decomp import SP_RECV, SP_SEND, SP_STORAGE, BrickDecomp, Exchanger, XPlan, exchange_loopback
from .order import LIST_SKIP, decomp_lists, tiled_scattered
from .stencils import STENCILS, Zigzag, apply_stencil, stencil_iters


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


def strong_tables(d, l, r, rank, size, subdim, alias_local=True):
    """Host tables of rank `rank` owning subdomains [l, r) (strong/main.cu:188-395 at run granularity): (copies, msgs, peers,
    alias).  copies / msgs: rows for brk_xplan_build_tables -- same-rank ghost runs as storage->storage links (unless aliased),
    the others packed per peer rank (canonical order on both sides: receiver Z-Morton id, run index), ONE send + ONE recv per
    peer.  alias[b] = the brick that holds brick b's data: identity, except (alias_local) the ghost bricks whose source subdomain
    lives on this rank."""
    nb, esz, nsub, allsubs = d.nbricks, 8, r - l, subdim ** 3
    off = lambda m, pos: (m * nb + pos) * 512 * esz          # byte offset inside storages[0]
    alias = np.arange(nsub * nb, dtype=np.uint32)
    copies = []
    recv, send = {}, {}       # peer -> list of (key=(receiver zmort, run), sub index, run)
    for m in range(nsub):
        z = l + m
        for q, (g, s) in enumerate(zip(d.ghost, d.skin)):
            # receive side: ghost run q of z comes from the subdomain in direction g.neighbor
            zs, owner, slot = neighbor_of(g["neighbor"], z, subdim, allsubs, size)
            if owner == rank and alias_local:
                alias[m * nb + g["pos"]:m * nb + g["pos"] + g["len"]] = np.arange(slot * nb + s["pos"], slot * nb + s["pos"] + g["len"],
                                                                                    dtype=np.uint32)
            elif owner == rank:
                copies.append(dict(stage=0, phase=0, src_space=SP_STORAGE, dst_space=SP_STORAGE, src_peer=-1, dst_peer=-1,
                                   src_off=off(slot, s["pos"]), dst_off=off(m, g["pos"]), bytes=g["len"] * 512 * esz))
            else:
                recv.setdefault(owner, []).append(((z, q), m, q))
            # send side: skin run q of z goes to the subdomain in direction s.neighbor
            zr, owner_r, _ = neighbor_of(s["neighbor"], z, subdim, allsubs, size)
            if owner_r != rank:
                send.setdefault(owner_r, []).append(((zr, q), m, q))
    peers = sorted(set(recv) | set(send))
    msgs = []
    for p in peers:
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
    return copies, msgs, peers, alias


def strong_lists(d, nsub, alias, rows_per_block=8, kpair=False):
    """(adjacency, list_all, list_domain) of a rank's nsub subdomains in one storage: block-diagonal adjacency (brick ids of
    subdomain m shifted by m*nbricks) with every aliased ghost brick replaced by its source, the sweep list over subdomains +
    ghost rings without the aliased ghost bricks (one tiled list per pattern of local neighbours), and the list of the subdomains
    themselves.  The lists stay subdomain after subdomain: measured on config 5 (1024^3 in 128^3 subdomains, one GPU, 13-point
    star) 13.1 ms per step, against 13.4 (8-row blocks) / 13.8 (4-row blocks) for ONE tiled sweep of the rank's bricks in the global
    frame -- rows of 128 bricks share more halos through LDS, but a subdomain is also a 16 x 16 x 16 cache tile whose faces come
    back from L2, and that is worth more.
    kpair: both lists in the k-paired order (order.tiled(..., kpair=True)) for sweeps of the 13- / 25-point stars through
    brk_stencil_apply_list_kpaired (two bricks per wave, k_star_km); a subdomain with aliased ghost faces is the k-paired order of its
    bounding box with LIST_SKIP in the holes."""
    nb = d.nbricks
    adj = np.concatenate([d.adj.astype(np.uint64) + m * nb for m in range(nsub)]).astype(np.uint32)
    lists = decomp_lists(d, rows_per_block, kpair=kpair)
    shift = lambda ids, m: np.where(ids == LIST_SKIP, ids, ids + np.uint32(m * nb)).astype(np.uint32)
    own = (alias == np.arange(alias.size, dtype=np.uint32)).reshape(nsub, nb)
    g3 = np.asarray(d.grid).astype(np.uint32)                    # [k][j][i] of the ghosted grid -> local brick id
    kk, jj, ii = (c.reshape(-1) for c in np.meshgrid(*(np.arange(n) for n in g3.shape), indexing="ij"))
    loc = g3.reshape(-1)
    cache, parts = {}, []
    for m in range(nsub):
        key = own[m].tobytes()
        if key not in cache:
            keep = own[m][loc]
            cache[key] = lists["all"] if keep.all() else tiled_scattered(loc[keep], ii[keep], jj[keep], kk[keep], rows_per_block, kpair=kpair)
        parts.append(shift(cache[key], m))
    return alias[adj], np.concatenate(parts), np.concatenate([shift(lists["domain"], m) for m in range(nsub)])


class StrongRank:
    def __init__(self, dom, sdom, rank, size, device, rows_per_block=8, alias_local=True, kpair=True):
        assert dom % sdom == 0 and (dom // sdom) & (dom // sdom - 1) == 0, "blocks per axis must be a power of two"
        self.dom, self.sdom, self.rank, self.size, self.device = dom, sdom, rank, size, torch.device(device)
        self.subdim = dom // sdom
        self.allsubs = self.subdim ** 3
        self.l, self.r = partition(self.allsubs, size, rank)
        self.nsub = self.r - self.l
        d = self.d = BrickDecomp((sdom, sdom, sdom), 8)        # ONE decomposition shared by all subdomains (strong/main.cpp:91-94)
        nb = self.nb = d.nbricks
        self.alias_local = bool(alias_local)
        self.kpair = bool(kpair)
        copies, msgs, self.peers, self.alias = strong_tables(d, self.l, self.r, rank, size, self.subdim, self.alias_local)
        adj, list_all, list_domain = strong_lists(d, self.nsub, self.alias, rows_per_block)
        self.rows_per_block = rows_per_block
        self._paired = None      # (list_all, list_domain, complete fraction) in the k-paired order, built on the first sweep of a star stencil
        self.swept_kpaired = False
        self.info = BrickInfo(adj, self.device)
        self.storages = [BrickStorage.allocate(self.nsub * nb, 512, self.device) for _ in range(3)]
        self.bricks = [Brick(self.info, st, 0) for st in self.storages]
        self.swept_bricks = int((list_all != LIST_SKIP).sum())          # per sweep over list_all
        self.list_all = grid_to_device(list_all, self.device)
        self.list_domain = grid_to_device(list_domain, self.device)
        self.grid_dev = grid_to_device(d.grid.reshape(-1), self.device)
        plan = XPlan.from_tables(self.rank, copies, msgs)
        # the view: buffers + device link tables inside libbrk.so; begin / transport / end are C-ABI calls
        self.xch = Exchanger(d, self.storages[0], self.rank, mode="packed", plan=plan)
        self.send_buf, self.recv_buf = self.xch.send_buf, self.xch.recv_buf
        self.bytes_remote, self.bytes_local = plan.bytes_remote, plan.bytes_local

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

    def sweeps(self, stencil, coeff=None, stream=None, zigzag=True):
        """strong/main.cpp:359-365: 1<-0; (2<-1; 1<-2) x (S/2-1); 0<-1 without the outer brick ring.  zigzag: successive sweeps walk
        the lists in alternating directions (a sweep starts with what the one before wrote last), see drivers.weak_cycle."""
        S = stencil_iters(stencil)
        b = self.bricks
        zz = Zigzag()
        rev = lambda: zigzag and zz.turn()
        # the 13- and 25-point stars sweep the same bricks in the k-paired order (two bricks per wave, k_star_km: round 5; the
        # results do not depend on the order); every other stencil keeps the plain tiled lists
        # -- when at least 90 % of the swept bricks lie in complete groups (4 i-neighbours + their 4 +k neighbours: a subdomain whose
        # ghost faces are all aliased is a 16^3 box, all complete; with 8 ranks x 64 subdomains of 128^3 the kept ghost faces make the
        # extents 17 / 18 and only 77 % complete -- the march kernel computes incomplete groups one brick at a time, which costs more
        # than the pairing gains)
        star = STENCILS.get(stencil, stencil) in (STENCILS["mpi13pt"], STENCILS["mpi25pt"])      # name or id
        kp = self.kpair and star and self.paired_lists()[2] >= 0.9
        la, ld = self.paired_lists()[:2] if kp else (self.list_all, self.list_domain)
        self.swept_kpaired = bool(kp)
        apply_stencil(stencil, b[0], b[1], ids=la, coeff=coeff, stream=stream, reverse=rev(), kpaired=kp)
        for _ in range(S // 2 - 1):
            apply_stencil(stencil, b[1], b[2], ids=la, coeff=coeff, stream=stream, reverse=rev(), kpaired=kp)
            apply_stencil(stencil, b[2], b[1], ids=la, coeff=coeff, stream=stream, reverse=rev(), kpaired=kp)
        apply_stencil(stencil, b[1], b[0], ids=ld, coeff=coeff, stream=stream, reverse=rev(), kpaired=kp)
        return S

    def paired_lists(self):
        """(list_all, list_domain, fraction of list_all's bricks in complete groups of 8) in the k-paired order, lists on the device;
        same bricks as the plain lists (checked)."""
        if self._paired is None:
            _, la, ld = strong_lists(self.d, self.nsub, self.alias, self.rows_per_block, kpair=True)
            assert int((la != LIST_SKIP).sum()) == self.swept_bricks
            complete = int((la.reshape(-1, 8) != LIST_SKIP).all(axis=1).sum()) * 8
            self._paired = (grid_to_device(la, self.device), grid_to_device(ld, self.device), complete / max(self.swept_bricks, 1))
        return self._paired


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
