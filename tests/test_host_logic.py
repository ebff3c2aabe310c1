"""CPU-side checks of the product's host logic (libbrk.so layout builders) against the pinned oracle and the
golden fixtures, plus: the C-ABI library loads and exports every symbol include/brk.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import bricklib_amd as bl
from bricklib_amd._lib import ABI_SYMBOLS, LIB_PATH, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "brk.h")).read()
    declared = set(re.findall(r"\b(brk_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(ABI_SYMBOLS), declared ^ set(ABI_SYMBOLS)
    L = C.CDLL(LIB_PATH)
    for s in ABI_SYMBOLS:
        assert hasattr(L, s), s
    assert lib.brk_abi_version() == 1013
    # development knobs stay out of the product library (they exist in `make DEV=1` / `VARIANTS=1` builds only)
    if os.environ.get("BRK_DEV_BUILD") != "1":
        for s in ("brk_set_variant", "brk_debug_pad_lds", "brk_debug_diag", "brk_debug_stamps"):
            assert not hasattr(L, s), s


def test_error_convention():
    rc = lib.brk_stencil_apply_range(99, None, None, None, 512, 0, 1, None, None)
    assert rc == -1 and b"stencil" in lib.brk_last_error()
    with pytest.raises(bl.BrkError):
        bl.BrickDecomp((20, 16, 16))          # not a multiple of the brick edge
    assert lib.brk_stencil_radius(bl.BRK_MPI25) == 4 and lib.brk_stencil_iters(bl.BRK_MPI13) == 4
    assert lib.brk_stencil_ncoeff(bl.BRK_MPI125) == 10


@pytest.mark.parametrize("dims", [(4, 4, 4), (6, 5, 4)])
def test_init_grid_golden(golden, dims):
    z = golden("init_grid_%dx%dx%d" % dims)
    grid, info = bl.init_grid(dims)
    assert (grid == z["grid"]).all() and (info.adj_host == z["adj"]).all()


@pytest.mark.parametrize("dims", [(16, 16, 16), (32, 32, 32), (24, 32, 40), (64, 64, 64)])
def test_decomp_golden(golden, dims):
    z = golden("decomp_%dx%dx%d" % dims)
    d = bl.BrickDecomp(dims, 8)
    assert d.sep_pos == [int(x) for x in z["sep_pos"]] and d.nbricks == int(z["nbricks"][0])
    assert (d.grid.reshape(-1) == z["grid"]).all()
    assert (d.adj[1:] == z["adj_from1"]).all() and (d.adj[0] == 0).all()
    assert (np.array(d.skin_size) == z["skin_size"]).all()
    assert (d.runs_table("ghost") == z["ghost"]).all() and (d.runs_table("skin") == z["skin"]).all()


@pytest.mark.parametrize("order", ["normal", "bad"])
@pytest.mark.parametrize("dims", [(16, 16, 16), (24, 32, 40), (64, 64, 64)])
def test_decomp_golden_other_section_orders(golden, dims, order):
    """brk_decomp_build_order with skin3d_normal / skin3d_bad against the reference's own tables; a list that misses a section,
    repeats one or names something that is not a section is refused"""
    z = golden("decomp_%s_%dx%dx%d" % ((order,) + dims))
    assert (np.array(bl.decomp.skin3d(order), dtype=np.uint64) == z["skinlist"]).all()
    d = bl.BrickDecomp(dims, 8, order=order)
    assert d.sep_pos == [int(x) for x in z["sep_pos"]] and d.nbricks == int(z["nbricks"][0])
    assert (d.grid.reshape(-1) == z["grid"]).all()
    assert (d.adj[1:] == z["adj_from1"]).all() and (d.adj[0] == 0).all()
    assert (np.array(d.skin_size) == z["skin_size"]).all()
    assert (d.runs_table("ghost") == z["ghost"]).all() and (d.runs_table("skin") == z["skin"]).all()
    good = bl.decomp.skin3d("good")
    for broken in (good[:-1], good + [good[0]], good[:-1] + [good[-1] | good[0] | (1 << 40)]):
        with pytest.raises(bl.BrkError):
            bl.BrickDecomp(dims, 8, order=broken)
    # a list of one's own: skin3d_good reversed, with separators sprinkled in
    mine = list(reversed(good))
    mine.insert(5, 0)
    d2 = bl.BrickDecomp(dims, 8, order=mine)
    assert d2.nbricks == int(z["nbricks"][0]) and sorted(np.unique(d2.grid).tolist()) == list(range(1, d2.nbricks))


def test_decomp_512_matches_survey_appendix_a(oracle):
    d = bl.BrickDecomp((512, 512, 512), 8)
    o = oracle.decomp((512, 512, 512))
    assert d.sep_pos == [238329, 262145, 287497] and len(d.ghost) == 42
    assert (d.grid.reshape(-1) == o.grid).all() and (d.adj == o.adj).all()
    assert sum(g["len"] for g in d.ghost) * 4096 == 103841792


def test_neighbors_rank_maps_dims(oracle, golden):
    z = golden("bitset_zmort")
    assert bl.allneighbors() == [int(x) for x in z["allneighbors"]]
    d = bl.BrickDecomp((16, 16, 16))
    assert d.skinlist == [int(x) for x in z["skin3d_good"]]
    for size in (1, 2, 4, 8, 6, 12):
        cart = bl.dims_create(size)
        assert cart[0] * cart[1] * cart[2] == size and cart[0] >= cart[1] >= cart[2]
        for rank in range(size):
            coo = bl.decomp.rank_to_coo(rank, cart)
            rm = bl.populate(d, cart, coo)
            want = oracle.populate_rank_map(cart, coo)
            assert [rm[n] for n in bl.allneighbors()] == [int(x) for x in want]
            assert rm[0] == rank
    assert bl.dims_create(8) == [2, 2, 2] and bl.dims_create(4) == [2, 2, 1] and bl.dims_create(2) == [2, 1, 1]


def test_rank_map_symmetry():
    """receiver/sender agreement needed by the exchange: map_R[n] == S  <=>  map_S[!n] == R"""
    d = bl.BrickDecomp((16, 16, 16))
    neg = {g["neighbor"]: s["neighbor"] for g, s in zip(d.ghost, d.skin)}
    for size in (2, 4, 8, 12):
        cart = bl.dims_create(size)
        maps = [dict(bl.populate(d, cart, bl.decomp.rank_to_coo(r, cart))) for r in range(size)]
        for r in range(size):
            for n, nn in neg.items():
                assert maps[maps[r][n]][nn] == r


def test_zmort_and_strong_partition(oracle, golden):
    z = golden("bitset_zmort")
    for (zid, d, p), g, s in zip(z["zq"], z["zget"], z["zset"]):
        assert lib.brk_zmort_get(int(zid), int(d)) == int(g)
        assert lib.brk_zmort_set(int(zid), int(d), int(p)) == int(s)
    for subdim, size in ((2, 4), (4, 8), (4, 3), (8, 8), (4, 5)):
        allsubs = subdim ** 3
        cover = []
        for rank in range(size):
            lr = (C.c_uint64 * 2)()
            lib.brk_strong_partition(allsubs, size, rank, lr)
            want = oracle.strong_partition(allsubs, size, rank)
            assert [int(lr[0]), int(lr[1])] == want[:2]
            cover += list(range(int(lr[0]), int(lr[1])))
        assert cover == list(range(allsubs))
        for n in bl.allneighbors():
            for zid in range(0, allsubs, 3):
                o, s = C.c_int(), C.c_int()
                nz = lib.brk_strong_neighbor(n, zid, subdim, allsubs, size, C.byref(o), C.byref(s))
                assert (int(nz), o.value, s.value) == oracle.strong_getrank(n, zid, subdim, size)


def test_populate_and_dims_match_reference_under_mpi(oracle, golden):
    """tests/golden/populate_ranks.npz + dims_create.npz = the REFERENCE's populate() (include/brick-mpi.h:730-753) and
    MPI_Dims_create as weak/args.cpp:105 calls it, run under mpirun -np {1,2,3,4,6,8,12,16} (oracle/gen_golden_mpi.py):
    the C ABI (brk_populate_rank_map, brk_dims_create) and the oracle restatement reproduce every rank's map."""
    z, dm = golden("populate_ranks"), golden("dims_create")["n_dims"]
    def smooth7(n):                       # brk_dims_create covers sizes whose prime factors are 2, 3, 5, 7 (include/brk.h)
        for p in (2, 3, 5, 7):
            while n % p == 0:
                n //= p
        return n == 1
    for n, d0, d1, d2 in dm:
        if smooth7(int(n)):
            assert bl.dims_create(int(n)) == [int(d0), int(d1), int(d2)], n
    d = bl.BrickDecomp((16, 16, 16))
    order = bl.allneighbors()
    for n in z["sizes"]:
        cart = [int(x) for x in z[f"dims_{n}"]]
        assert bl.dims_create(int(n)) == cart
        for rank in range(int(n)):
            coo = [int(x) for x in z[f"coo_{n}"][rank]]
            assert bl.decomp.rank_to_coo(rank, cart) == coo                      # MPI_Cart_coords is row-major
            rm = bl.populate(d, cart, coo)
            want = [int(x) for x in z[f"map_{n}"][rank]]
            assert [rm[s] for s in order] == want, (n, rank)
            assert [int(x) for x in oracle.populate_rank_map(cart, coo)] == want


def test_strong_ownership_matches_reference_under_mpi(oracle, golden):
    """tests/golden/strong_getrank.npz = getrank() and the ownership ranges of the reference's strong/args.cpp:36-56,104-113,
    compiled as they are and run under mpirun (9 cases, 1..8 ranks, 8..512 subdomains): brk_strong_partition /
    brk_strong_neighbor and the oracle restatement agree for every (subdomain, neighbour)."""
    z = golden("strong_getrank")
    for n, d, s in z["cases"]:
        n, d, s = int(n), int(d), int(s)
        subdim = d // s
        allsubs = subdim ** 3
        rng = z[f"range_{n}_{d}_{s}"]
        for rank in range(n):
            lr = (C.c_uint64 * 2)()
            lib.brk_strong_partition(allsubs, n, rank, lr)
            assert [int(lr[0]), int(lr[1])] == [int(rng[rank][0]), int(rng[rank][1])]
            assert oracle.strong_partition(allsubs, n, rank)[:2] == [int(rng[rank][0]), int(rng[rank][1])]
        g = z[f"getrank_{n}_{d}_{s}"]
        assert g.shape[0] == allsubs * 27
        for owner, zid, nset, znew, dst, sub in g:
            o, sl = C.c_int(), C.c_int()
            nz = lib.brk_strong_neighbor(int(nset), int(zid), subdim, allsubs, n, C.byref(o), C.byref(sl))
            assert (int(nz), o.value, sl.value) == (int(znew), int(dst), int(sub)), (n, d, s, int(zid), int(nset))
            assert int(rng[int(owner)][0]) <= int(zid) < int(rng[int(owner)][1])
        for owner, zid, nset, znew, dst, sub in g[::7]:
            assert oracle.strong_getrank(int(nset), int(zid), subdim, n) == (int(znew), int(dst), int(sub))


def test_exchange_plans_host_side(oracle):
    """brk_xplan_*: plans are host bookkeeping (no GPU): every ghost brick of every rank is written exactly once by the copies
    of a packed / direct / shift plan, sends and receives of all ranks pair up message by message, error paths report."""
    from bricklib_amd.decomp import SP_RECV, SP_SEND, SP_STORAGE, XPlan, populate, rank_to_coo
    dom = (32, 16, 24)
    for world in (1, 2, 3, 8):
        cart = bl.dims_create(world)
        for mode in ("packed", "direct", "shift"):
            plans = []
            for r in range(world):
                d = bl.BrickDecomp(dom)
                populate(d, cart, rank_to_coo(r, cart))
                plans.append((d, XPlan.from_decomp(d, r, mode, 512)))
            for r, (d, p) in enumerate(plans):
                assert p.my_rank == r and p.nstages == (3 if mode == "shift" else 1)
                # bytes landing in the storage: each ghost brick exactly once over all stages (copies into storage + direct receives)
                hits = np.zeros(d.nbricks, dtype=np.int32)
                for c in p.copies:
                    if c["dst_space"] == SP_STORAGE:
                        assert c["dst_off"] % 4096 == 0 and c["bytes"] % 4096 == 0
                        hits[c["dst_off"] // 4096:(c["dst_off"] + c["bytes"]) // 4096] += 1
                for m in p.msgs:
                    if not m["is_send"] and m["space"] == SP_STORAGE:
                        hits[m["off"] // 4096:(m["off"] + m["bytes"]) // 4096] += 1
                assert (hits[d.sep_pos[1]:d.sep_pos[2]] == 1).all() and hits[:d.sep_pos[1]].sum() == 0, (world, mode, r)
                # every send has its receive on the peer, in order, with the same size
                for st in range(p.nstages):
                    for peer in p.peers(st):
                        mine = [m["bytes"] for m in p.msgs if m["stage"] == st and m["is_send"] and m["peer"] == peer]
                        theirs = [m["bytes"] for m in plans[peer][1].msgs if m["stage"] == st and not m["is_send"] and m["peer"] == r]
                        assert mine == theirs, (world, mode, r, peer, st)
                        if mode != "direct":
                            assert len(mine) <= 1 and p.buffer_bytes(st, peer, SP_SEND) == sum(mine)
                            assert plans[peer][1].buffer_bytes(st, r, SP_RECV) == sum(mine)
                if mode == "packed":
                    assert len(p.peers()) <= 7
                if mode == "shift":
                    assert sum(1 for m in p.msgs if m["is_send"]) <= 6
    # error paths
    d = bl.BrickDecomp(dom)
    populate(d, [1, 1, 1], [0, 0, 0])
    ranks = (C.c_int * 27)(*[0] * 27)
    h = C.c_void_p()
    assert lib.brk_xplan_build_decomp(d._h, ranks, 0, 7, 0, 512, C.byref(h)) != 0 and b"mode" in lib.brk_last_error()
    assert lib.brk_xplan_build_decomp(d._h, ranks, -1, 0, 0, 512, C.byref(h)) != 0
    assert lib.brk_xplan_build_decomp(d._h, ranks, 0, 0, 0, 100, C.byref(h)) != 0
    bad = dict(stage=0, phase=0, src_space=SP_STORAGE, dst_space=SP_SEND, src_peer=-1, dst_peer=1, src_off=8, dst_off=0, bytes=4096)
    with pytest.raises(bl.BrkError, match="multiples of 16"):
        XPlan.from_tables(0, [bad], [])
    with pytest.raises(bl.BrkError, match="receive buffer"):
        XPlan.from_tables(0, [dict(bad, src_off=0, dst_space=SP_RECV)], [])
    with pytest.raises(bl.BrkError, match="sends leave from send buffers"):
        XPlan.from_tables(0, [], [dict(stage=0, is_send=1, peer=1, space=SP_RECV, off=0, bytes=4096)])
    ok = XPlan.from_tables(2, [dict(bad, src_off=0)], [dict(stage=0, is_send=1, peer=1, space=SP_SEND, off=0, bytes=4096)])
    assert ok.bytes_remote == 4096 and ok.buffer_bytes(0, 1, SP_SEND) == 4096 and ok.my_rank == 2


@pytest.mark.parametrize("dom", [(16, 16, 16), (32, 16, 24), (24, 40, 16)])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 6, 8, 12])
def test_exchange_plans_route_the_right_bytes(oracle, dom, world):
    """The plans of ALL ranks of a periodic grid, interpreted in one address space (numpy; no GPU, no processes): begin copies,
    messages matched in order per (stage, sender, receiver), end copies -- stage by stage, for the packed, direct and shift
    forms -- must leave every rank's storage exactly as the oracle's run-by-run exchange (brick-mpi.h:466-495) does.  This is the
    N > 1 data path of bench.py minus the wire: the same tables drive brk_exchange_begin / _transport / _end on the GPUs."""
    from bricklib_amd.decomp import SP_RECV, SP_SEND, SP_STORAGE, XPlan, populate, rank_to_coo
    cart = bl.dims_create(world)
    od = oracle.decomp(dom)
    nbytes = od.nbricks * 4096
    init = [(np.arange(od.nbricks * 512, dtype=np.float64) + r * 1e9) for r in range(world)]
    maps = [oracle.populate_rank_map(cart, rank_to_coo(r, cart)) for r in range(world)]
    want = [a.copy() for a in init]
    oracle.exchange(od, want, maps, 512)
    for mode in ("packed", "direct", "shift"):
        plans = []
        for r in range(world):
            d = bl.BrickDecomp(dom)
            populate(d, cart, rank_to_coo(r, cart))
            plans.append(XPlan.from_decomp(d, r, mode, 512))
        st = [a.copy().view(np.uint8) for a in init]
        assert all(s_.size == nbytes for s_ in st)
        for stage in range(plans[0].nstages):
            buf = {}     # (rank, space, peer) -> bytearray-like

            def space(r, sp, peer, need):
                if sp == SP_STORAGE:
                    return st[r]
                b = buf.setdefault((r, sp, peer), np.zeros(plans[r].buffer_bytes(stage, peer, sp), dtype=np.uint8))
                assert need <= b.size, (mode, r, sp, peer, need, b.size)
                return b

            def run_copies(phase):
                for r, p in enumerate(plans):
                    for c in p.copies:
                        if c["stage"] == stage and c["phase"] == phase:
                            src = space(r, c["src_space"], c["src_peer"], c["src_off"] + c["bytes"])
                            dst = space(r, c["dst_space"], c["dst_peer"], c["dst_off"] + c["bytes"])
                            dst[c["dst_off"]:c["dst_off"] + c["bytes"]] = src[c["src_off"]:c["src_off"] + c["bytes"]].copy()

            run_copies(0)
            for r, p in enumerate(plans):
                for peer in range(world):
                    sends = [m for m in p.msgs if m["stage"] == stage and m["is_send"] and m["peer"] == peer]
                    recvs = [m for m in plans[peer].msgs if m["stage"] == stage and not m["is_send"] and m["peer"] == r]
                    assert [m["bytes"] for m in sends] == [m["bytes"] for m in recvs], (mode, stage, r, peer)
                    for a, b in zip(sends, recvs):
                        src = space(r, a["space"], peer, a["off"] + a["bytes"])
                        dst = space(peer, b["space"], r, b["off"] + b["bytes"])
                        dst[b["off"]:b["off"] + b["bytes"]] = src[a["off"]:a["off"] + a["bytes"]].copy()
            run_copies(1)
        for r in range(world):
            assert (st[r].view(np.float64) == want[r]).all(), (mode, r)


def test_exchange_plan_covers_ghost_once():
    d = bl.BrickDecomp((32, 32, 32))
    covered = np.zeros(d.nbricks, dtype=np.int32)
    for g in d.ghost:
        covered[g["pos"]:g["pos"] + g["len"]] += 1
    assert (covered[d.sep_pos[1]:d.sep_pos[2]] == 1).all() and covered[:d.sep_pos[1]].sum() == 0
    for s in d.skin:
        assert d.sep_pos[0] <= s["pos"] and s["pos"] + s["len"] <= d.sep_pos[1]


@pytest.mark.parametrize("dom,sweeps", [((64, 48, 56), 8), ((40, 32, 48), 4), ((16, 16, 16), 2), ((96, 96, 96), 8)])
def test_deep_schedule_dataflow(dom, sweeps):
    """drivers.weak_cycle_deep replayed on version stamps: every pass of sweep t reads only sweep t-1 data, the early
    passes touch nothing the exchange reads or writes, and each sweep covers exactly the reference's brick set
    (weak/main.cpp:186-207: all bricks incl. ghost, the last sweep without the ghost ring)."""
    from bricklib_amd.order import LIST_SKIP, trapezoid_lists
    d = bl.BrickDecomp(dom)
    inner, shell = trapezoid_lists(d, sweeps)
    sep0, sep1, sep2 = d.sep_pos
    nb = d.nbricks
    GARBAGE = -100
    ver = {"A": np.zeros(nb, dtype=np.int64), "B": np.full(nb, GARBAGE, dtype=np.int64)}
    ver["A"][sep1:sep2] = GARBAGE            # ghosts of A are stale until the exchange lands
    adj = d.adj
    # L(b): bricks from the outside of the ghosted grid
    T = d.grid.shape
    kk, jj, ii = np.meshgrid(*[np.minimum(np.arange(n), n - 1 - np.arange(n)) for n in T], indexing="ij")
    L = np.zeros(nb, dtype=np.int64)
    L[d.grid.reshape(-1)] = np.minimum(np.minimum(kk, jj), ii).reshape(-1)

    def run(ids, t, early):
        src, dst = ("A", "B") if t % 2 == 1 else ("B", "A")
        ids = ids[ids != LIST_SKIP].astype(np.int64)
        if ids.size == 0:
            return ids
        nbrs = adj[ids]                                           # [n][27]; 0 = outside the ghosted grid
        need = ver[src][nbrs]
        if early:
            assert (L[ids] >= t + 1).all()
            assert (need == t - 1).all()                          # nothing stale, nothing from a later sweep
            if src == "A":
                assert (L[nbrs] >= 1).all() and (nbrs != 0).all()  # never reads a ghost brick the exchange is writing
            if dst == "A":
                assert (L[ids] >= 2).all()                        # never writes what the exchange is sending
        else:
            # late passes: everything a DOMAIN brick will ever depend on is current; stale data may only enter
            # through the outside of the ghosted grid (brick 0), as in the reference
            assert ((need == t - 1) | (nbrs == 0)).all()
        ver[dst][ids] = t
        return ids

    for t in range(1, sweeps + 1):
        run(inner[t - 1], t, True)
    ver["A"][np.where(ver["A"][sep1:sep2] == GARBAGE)[0] + sep1] = 0      # exchange complete: A's ghosts are sweep-0 data
    assert (ver["A"][sep1:sep2] == 0).all()
    for t in range(1, sweeps + 1):
        early_ids = inner[t - 1][inner[t - 1] != LIST_SKIP]
        late_ids = run(shell[t - 1], t, False)
        both = np.sort(np.concatenate([early_ids.astype(np.int64), late_ids]))
        assert np.array_equal(both, np.arange(1, sep1 if t == sweeps else sep2))
    assert sweeps % 2 == 0 and (ver["A"][1:sep1] == sweeps).all()


@pytest.mark.parametrize("dom,sweeps", [((64, 48, 56), 8), ((64, 64, 64), 4), ((32, 32, 32), 8)])
def test_deep_schedule_early_start_split(dom, sweeps):
    """round 5, weak_cycle_deep(state=...): the late pass of the LAST sweep is cut into (surface, rest) so that the next exchange can
    start under `rest`.  Held here: the two parts are exactly that pass's brick set; the surface part is [sep0, sep1) -- what the
    exchange sends; `rest` writes only bricks two and more inside the ghosted grid (the exchange reads L = 1 of A and writes L = 0 of
    A) and -- the last sweep being an even one -- reads B, not A."""
    from bricklib_amd.order import LIST_SKIP, last_shell_split, trapezoid_lists
    d = bl.BrickDecomp(dom)
    _, shell = trapezoid_lists(d, sweeps)
    surf, rest = last_shell_split(d, sweeps)
    real = lambda x: np.sort(x[x != LIST_SKIP].astype(np.int64))
    assert np.array_equal(real(np.concatenate([surf, rest])), real(shell[sweeps - 1]))
    assert np.array_equal(real(surf), np.arange(d.sep_pos[0], d.sep_pos[1]))
    T = d.grid.shape
    kk, jj, ii = np.meshgrid(*[np.minimum(np.arange(n), n - 1 - np.arange(n)) for n in T], indexing="ij")
    L = np.zeros(d.nbricks, dtype=np.int64)
    L[d.grid.reshape(-1)] = np.minimum(np.minimum(kk, jj), ii).reshape(-1)
    assert (L[real(rest)] >= 2).all() and (L[real(surf)] == 1).all()
    assert sweeps % 2 == 0           # sweep `sweeps` reads B and writes A (weak_cycle_deep's ping-pong)


def test_host_bookkeeping_under_address_and_ub_sanitizers(tmp_path):
    """brk_host.cpp + brk_xplan.cpp (everything in libbrk.so that is plain C++: init_grid, BrickDecomp tables, rank maps,
    exchange plans in all modes for 1..27 ranks, strong partition / getrank / Z-Morton, error paths) built with
    -fsanitize=address,undefined and driven by tests/probe_asan_host.cpp; no report, no leak.  (GPU sanitizers are not available
    on the test pool.)"""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    csrc = os.path.join(ROOT, "bricklib_amd", "csrc")
    flags = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"), "-I", csrc]
    lib_ = tmp_path / "libbrkhost_asan.so"
    r = subprocess.run(["g++", *flags, "-fPIC", "-shared", os.path.join(csrc, "brk_host.cpp"), os.path.join(csrc, "brk_xplan.cpp"), "-o", str(lib_)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0 and b"asan" in r.stdout.lower():
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stdout.decode()
    exe = tmp_path / "probe_asan_host"
    subprocess.check_call(["g++", *flags, os.path.join(ROOT, "tests", "probe_asan_host.cpp"), "-o", str(exe), "-L", str(tmp_path), "-lbrkhost_asan",
                           "-Wl,-rpath," + str(tmp_path)])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0 and "probe_asan_host: ok" in out and "ERROR" not in out and "runtime error" not in out, out


@pytest.mark.parametrize("dims", [(64, 64, 64), (32, 48, 64), (24, 32, 40)])
def test_kpaired_sweep_lists_hold_the_same_bricks(dims):
    """order.decomp_lists / trapezoid_lists with kpair=True (the order brk_stencil_apply_list_kpaired wants): the same brick sets as
    the plain tiled lists, every list a whole number of 8-entry groups, and inside the full boxes entry w+4 of a group is the +k
    neighbour of entry w"""
    from bricklib_amd.decomp import BrickDecomp
    from bricklib_amd.order import LIST_SKIP, decomp_lists, trapezoid_lists
    d = BrickDecomp(dims, 8)
    a, b = decomp_lists(d), decomp_lists(d, kpair=True)
    for k in a:
        assert np.array_equal(np.sort(a[k][a[k] != LIST_SKIP]), np.sort(b[k][b[k] != LIST_SKIP])), k
        assert b[k].size % 8 == 0
    ia, sa = trapezoid_lists(d, 4)
    ib, sb = trapezoid_lists(d, 4, kpair=True)
    for u, v in zip(ia + sa, ib + sb):
        assert np.array_equal(np.sort(u[u != LIST_SKIP]), np.sort(v[v != LIST_SKIP]))
    g = b["all"].reshape(-1, 2, 4)
    both = (g[:, 0, :] != LIST_SKIP) & (g[:, 1, :] != LIST_SKIP)
    up = d.adj.reshape(-1, 27)[g[:, 0, :][both], 13 + 9]
    assert both.any() and np.array_equal(up, g[:, 1, :][both])


def _strong_cycle_on_host(oracle, sid, dom, sdom, size, alias_local, seed=5):
    """the strong cycle of `size` ranks emulated from the host tables alone (strong.strong_tables / strong_lists): numpy storages,
    copies and messages executed literally, sweeps by the oracle over the lists.  Returns {zmort id: subdomain cells}."""
    from bricklib_amd.decomp import SP_RECV, SP_SEND, SP_STORAGE, BrickDecomp
    from bricklib_amd.order import LIST_SKIP
    from bricklib_amd.strong import partition, strong_lists, strong_tables, zmort_coords
    subdim = dom // sdom
    d = BrickDecomp((sdom,) * 3, 8)
    nb, ext = d.nbricks, (sdom + 16,) * 3
    grid = np.ascontiguousarray(d.grid.reshape(-1), dtype=np.uint32)
    ranks = []
    for rank in range(size):
        l, r = partition(subdim ** 3, size, rank)
        copies, msgs, peers, alias = strong_tables(d, l, r, rank, size, subdim, alias_local)
        adj, list_all, list_domain = strong_lists(d, r - l, alias, 8)
        st = [np.zeros((r - l) * nb * 512) for _ in range(3)]
        for m in range(r - l):
            o = [c * sdom for c in zmort_coords(l + m)]
            arr = oracle.fill_array_global(ext, [x - 8 for x in o], (dom,) * 3, seed)
            arr[:8], arr[-8:], arr[:, :8], arr[:, -8:], arr[:, :, :8], arr[:, :, -8:] = (-1e3,) * 6     # ghosts must come from the exchange
            oracle.copy_to_brick(ext, (0,) * 3, (0,) * 3, arr, grid, st[0][m * nb * 512:(m + 1) * nb * 512], 512)
        ranks.append(dict(l=l, r=r, copies=copies, msgs=msgs, adj=np.ascontiguousarray(adj), st=st, alias=alias,
                          all=list_all[list_all != LIST_SKIP], domain=list_domain[list_domain != LIST_SKIP], send={}, recv={}))
        if alias_local and size == 1:
            assert not copies and not msgs and np.array_equal(np.sort(ranks[-1]["all"]), np.sort(ranks[-1]["domain"]))
    S = bl.stencil_iters(sid)
    for _ in range(2):
        for R in ranks:                                               # links + pack (phase 0)
            for m in R["msgs"]:
                R["send" if m["is_send"] else "recv"][m["peer"]] = np.zeros(m["bytes"] // 8)
            a = R["st"][0]
            for c in R["copies"]:
                if c["phase"] == 0:
                    src = a[c["src_off"] // 8:(c["src_off"] + c["bytes"]) // 8]
                    dst = a if c["dst_space"] == SP_STORAGE else R["send"][c["dst_peer"]]
                    dst[c["dst_off"] // 8:(c["dst_off"] + c["bytes"]) // 8] = src.copy()
        for rk, R in enumerate(ranks):                                # transport
            for p, buf in R["send"].items():
                ranks[p]["recv"][rk][:] = buf
        for R in ranks:                                               # unpack (phase 1) + sweeps 1<-0; (2<-1; 1<-2)...; 0<-1
            a = R["st"][0]
            for c in R["copies"]:
                if c["phase"] == 1:
                    assert c["src_space"] == SP_RECV
                    a[c["dst_off"] // 8:(c["dst_off"] + c["bytes"]) // 8] = R["recv"][c["src_peer"]][c["src_off"] // 8:(c["src_off"] + c["bytes"]) // 8]
            seq = [(0, 1, "all")] + [(1, 2, "all"), (2, 1, "all")] * (S // 2 - 1) + [(1, 0, "domain")]
            for s_, t_, which in seq:
                oracle.brick_stencil(sid, R["adj"], R["st"][s_], R["st"][t_], 512, ids=np.ascontiguousarray(R[which]))
    out = {}
    for R in ranks:
        for m in range(R["r"] - R["l"]):
            got = np.zeros(ext[::-1])
            oracle.copy_from_brick((sdom,) * 3, (0,) * 3, (8,) * 3, got, grid, R["st"][0][m * nb * 512:(m + 1) * nb * 512], 512)
            out[R["l"] + m] = got[8:-8, 8:-8, 8:-8].copy()
    return out, sum(R["all"].size for R in ranks)


@pytest.mark.parametrize("size", [1, 2, 3, 8])
def test_strong_aliased_local_ghosts_give_identical_cells(oracle, size):
    """strong.py alias_local: ghost runs whose source subdomain lives on the same rank are reached through the adjacency table
    instead of being copied and swept.  Two cycles of the 13-point star over 32^3 in 16^3 subdomains on 1, 2, 3 and 8 ranks, all on
    the host from the plan tables: BIT-identical subdomain cells with and without aliasing, equal to the global periodic field,
    fewer bricks swept (none of the ghost ring on one rank, all of it on eight)."""
    from bricklib_amd.strong import zmort_coords
    sid, dom, sdom = 3, 32, 16                                        # BRK_MPI13
    plain, n_plain = _strong_cycle_on_host(oracle, sid, dom, sdom, size, False)
    alias, n_alias = _strong_cycle_on_host(oracle, sid, dom, sdom, size, True)
    assert sorted(plain) == sorted(alias) == list(range(8))
    for z in plain:
        assert np.array_equal(plain[z], alias[z]), z
    assert n_alias < n_plain if size < 8 else n_alias == n_plain
    if size == 1:
        assert n_alias == 8 * 8                                       # 2^3 bricks per subdomain, the ghost rings are gone
    gext = (dom + 16,) * 3
    cur = oracle.fill_array_global(gext, (-8,) * 3, (dom,) * 3, 5)
    S, Rr = bl.stencil_iters(sid), oracle.radius(sid)
    for _ in range(2):
        for s_ in range(1, S + 1):
            nxt = np.zeros_like(cur)
            oracle.array_stencil(sid, cur, nxt, (s_ * Rr,) * 3, tuple(g - s_ * Rr for g in gext))
            cur = nxt
        core = cur[8:-8, 8:-8, 8:-8].copy()
        cur = np.pad(core, 8, mode="wrap")
    for z, cells in alias.items():
        o = [c * sdom for c in zmort_coords(z)]
        assert np.abs(cells - core[o[2]:o[2] + sdom, o[1]:o[1] + sdom, o[0]:o[0] + sdom]).max() <= 1e-12


@pytest.mark.parametrize("shape", [(16, 16, 16), (6, 9, 7), (5, 66, 13), (4, 3, 20), (18, 18, 18)])
def test_tiled_order_from_coordinates_equals_tiled_order_of_the_box(shape):
    """order.tiled_scattered (bricks given by coordinates, any order, holes allowed) reproduces order.tiled exactly on a full box,
    and with holes keeps every kept brick once, in runs of i-neighbours separated by LIST_SKIP"""
    from bricklib_amd.order import LIST_SKIP, tiled, tiled_scattered
    nk, nj, ni = shape
    ids3 = np.arange(1, nk * nj * ni + 1, dtype=np.uint32).reshape(shape)
    kk, jj, ii = (c.reshape(-1) for c in np.meshgrid(np.arange(nk), np.arange(nj), np.arange(ni), indexing="ij"))
    rng = np.random.default_rng(0)
    perm = rng.permutation(ids3.size)
    assert np.array_equal(tiled_scattered(ids3.reshape(-1)[perm], ii[perm] + 5, jj[perm] - 3, kk[perm]), tiled(ids3))
    keep = rng.random(ids3.size) < 0.7
    got = tiled_scattered(ids3.reshape(-1)[keep], ii[keep], jj[keep], kk[keep])
    real = got[got != LIST_SKIP]
    assert got.size % 64 == 0 and np.array_equal(np.sort(real), ids3.reshape(-1)[keep])
    for a, b in zip(got[:-1], got[1:]):          # list neighbours are i-neighbours or separated by a LIST_SKIP / a run of 8n entries
        if a != LIST_SKIP and b != LIST_SKIP and b != a + 1:
            pos = int(np.nonzero(got == b)[0][0])
            run = 0
            while pos - 1 - run >= 0 and got[pos - 1 - run] != LIST_SKIP and (run == 0 or got[pos - run] == got[pos - 1 - run] + 1):
                run += 1
            assert run % 8 == 0, (a, b, run)
