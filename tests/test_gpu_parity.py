"""Parity of the HIP path (through the C ABI) against the pinned CPU oracle.  Needs an MI355X."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import bricklib_amd as bl
from bricklib_amd.brick import grid_to_device
from bricklib_amd.decomp import LinkTable, local_links, rank_origin, rank_to_coo
from bricklib_amd.drivers import weak_cycle
from oracle.oracle import ST_7PT_COEFF, ST_27PT_COEFF, ST_MPI7, ST_MPI13, ST_MPI25, ST_MPI125

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _set_variant_or_skip(lib, sid, variant):
    """the earlier kernel generations are compiled only with `make -C bricklib_amd/csrc VARIANTS=1` (tools/kbench A/B runs); the
    default build carries one kernel per stencil, and a variant test then has nothing to run"""
    if not hasattr(lib, "brk_set_variant"):
        if variant != 0:
            pytest.skip("kernel variants are not part of this build (make -C bricklib_amd/csrc VARIANTS=1)")
        return
    if variant != 0 and lib.brk_set_variant(sid, variant) != 0:
        pytest.skip("kernel variants are not part of this build (make -C bricklib_amd/csrc VARIANTS=1)")
    assert lib.brk_set_variant(sid, variant) == 0

TOL = 1e-12   # fp64, <=125 terms of magnitude <1: observed ~1e-15; north-star bar is 1e-10
NAMES = {"7pt": ST_7PT_COEFF, "27pt": ST_27PT_COEFF, "mpi7pt": ST_MPI7, "mpi13pt": ST_MPI13, "mpi25pt": ST_MPI25,
         "mpi125pt": ST_MPI125}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def single_setup(oracle, dev, N, step, seed=42):
    """single/ driver shape (stencils/3axis.cpp:14-30): lexicographic grid with a one-brick ghost ring."""
    STRIDE, STRIDEG, SB = N + 32, N + 16, (N + 16) // 8
    grid, bInfo = bl.init_grid((SB, SB, SB), dev)
    arr = oracle.fill_array_global((STRIDE,) * 3, (0, 0, 0), (STRIDE,) * 3, seed)
    st = bl.BrickStorage.allocate(bInfo.nbricks, step, dev)
    bIn = bl.Brick(bInfo, st, 0)
    grid_dev = grid_to_device(grid, dev)
    bl.copyToBrick((STRIDEG,) * 3, (8,) * 3, (0,) * 3, torch.from_numpy(arr).to(dev), grid_dev, bIn)
    g3 = grid.reshape(SB, SB, SB)
    ids = np.ascontiguousarray(g3[1:-1, 1:-1, 1:-1]).reshape(-1)
    return grid, grid_dev, bInfo, arr, st, bIn, ids


def test_copy_to_brick_matches_oracle(oracle, dev):
    N = 32
    grid, grid_dev, bInfo, arr, st, bIn, ids = single_setup(oracle, dev, N, 1024)
    want = np.zeros(bInfo.nbricks * 1024)
    oracle.copy_to_brick((N + 16,) * 3, (8,) * 3, (0,) * 3, arr, grid, want, 1024)
    assert (st.dat.cpu().numpy() == want).all()


def test_golden_single_N16(oracle, golden, dev):
    """Outputs the reference headers produced (tests/golden/single_N16.npz) for 7-pt and 27-pt."""
    z = golden("single_N16")
    grid, grid_dev, bInfo, arr, st, bIn, ids = single_setup(oracle, dev, 16, 1024)
    assert (ids == z["ids"]).all()
    bOut = bl.Brick(bInfo, st, 512)
    coeff = oracle.coeffs(129, 1)
    ids_dev = grid_to_device(ids, dev)
    for name, key in (("7pt", "out7"), ("27pt", "out27")):
        bl.apply_stencil(name, bIn, bOut, ids=ids_dev, coeff=coeff)
        got = st.dat.cpu().numpy().reshape(-1, 1024)[ids, 512:]
        assert np.abs(got - z[key]).max() <= TOL


@pytest.mark.parametrize("name", list(NAMES))
@pytest.mark.parametrize("N,step", [(16, 512), (40, 1024)])
def test_stencil_list_form(oracle, dev, name, N, step):
    grid, grid_dev, bInfo, arr, st, bIn, ids = single_setup(oracle, dev, N, step)
    if step == 1024:
        sto, bOut = st, bl.Brick(bInfo, st, 512)
    else:
        sto = bl.BrickStorage.allocate(bInfo.nbricks, step, dev)
        bOut = bl.Brick(bInfo, sto, 0)
    coeff = oracle.coeffs(129, 1) if name in ("7pt", "27pt") else None
    bl.apply_stencil(name, bIn, bOut, ids=grid_to_device(ids, dev), coeff=coeff)
    got = sto.dat.cpu().numpy().reshape(-1, step)[:, step - 512:]
    hin = np.ascontiguousarray(st.dat.cpu().numpy().reshape(-1, step)[:, :512]).reshape(-1)
    want = np.zeros_like(hin)
    oracle.brick_stencil(NAMES[name], bInfo.adj_host, hin, want, 512, ids=ids, coeff=coeff)
    err = np.abs(got[ids] - want.reshape(-1, 512)[ids]).max()
    assert err <= TOL, err
    # bricks outside the list are untouched
    mask = np.ones(bInfo.nbricks, bool)
    mask[ids] = False
    assert (got[mask] == (0.0 if step == 512 else 0.0)).all()
    # and the reference's own acceptance test (compareBrick, 1e-6) against the array sweep
    aout = np.zeros_like(arr)
    oracle.array_stencil(NAMES[name], arr, aout, (16,) * 3, (16 + N,) * 3, coeff=coeff)
    assert bl.compareBrick((N,) * 3, (8,) * 3, (8,) * 3, torch.from_numpy(aout).to(dev), grid_dev, bOut)


def test_ragged_list_lengths_and_empty(oracle, dev):
    """list lengths that are not multiples of the 4-brick workgroup / 8-block XCD group, incl. empty."""
    grid, grid_dev, bInfo, arr, st, bIn, ids = single_setup(oracle, dev, 24, 512)
    hin = st.dat.cpu().numpy().copy()
    coeff = oracle.coeffs(129, 1)
    for n in (0, 1, 3, 5, 26, 27):
        sto = bl.BrickStorage.allocate(bInfo.nbricks, 512, dev)
        bOut = bl.Brick(bInfo, sto, 0)
        sub = ids[:n]
        bl.apply_stencil("7pt", bIn, bOut, ids=grid_to_device(sub, dev) if n else torch.zeros(0, dtype=torch.int32, device=dev),
                         coeff=coeff)
        want = np.zeros_like(hin)
        if n:
            oracle.brick_stencil(ST_7PT_COEFF, bInfo.adj_host, hin, want, 512, ids=sub, coeff=coeff)
        assert np.abs(sto.dat.cpu().numpy() - want).max() <= TOL


@pytest.mark.parametrize("name", ["mpi7pt", "mpi13pt", "mpi25pt", "mpi125pt"])
def test_range_form_decomp_order(oracle, dev, name):
    """brick ranges over BrickDecomp numbering (weak/ shape): [1,sep2) sweep incl. ghost ring reading dummy brick 0."""
    dom = (24, 16, 32)
    d = bl.BrickDecomp(dom)
    info = d.getBrickInfo(dev)
    ext = tuple(x + 16 for x in dom)
    arr = oracle.fill_array_global(ext, (0, 0, 0), ext, 7)
    A, B = info.allocate(512, dev), info.allocate(512, dev)
    bA, bB = bl.Brick(info, A, 0), bl.Brick(info, B, 0)
    gdev = grid_to_device(d.grid.reshape(-1), dev)
    bl.copyToBrick(ext, (0,) * 3, (0,) * 3, torch.from_numpy(arr).to(dev), gdev, bA)
    bl.apply_stencil(name, bA, bB, brick_range=(1, d.sep_pos[2]))
    hin = A.dat.cpu().numpy()
    want = np.zeros_like(hin)
    oracle.brick_stencil(NAMES[name], d.adj, hin, want, 512, n=d.sep_pos[2] - 1, first=1)
    assert np.abs(B.dat.cpu().numpy() - want).max() <= TOL


def test_array_brick_roundtrip_and_compare(oracle, dev):
    """copy() of stencils/copy.cpp:60-68 + compareBrick verdicts incl. the tolerance rule."""
    N = 24
    grid, grid_dev, bInfo, arr, st, bIn, ids = single_setup(oracle, dev, N, 512)
    arr_dev = torch.from_numpy(arr).to(dev)
    back = torch.zeros_like(arr_dev)
    bl.copyFromBrick((N + 16,) * 3, (8,) * 3, (0,) * 3, back, grid_dev, bIn)
    inner = (slice(8, 8 + N + 16),) * 3
    assert torch.equal(back[inner], arr_dev[inner])
    assert bl.compareBrick((N,) * 3, (8,) * 3, (8,) * 3, arr_dev, grid_dev, bIn)
    for eps, want in ((1e-7, True), (1e-5, False)):
        a2 = arr_dev.clone()
        a2[16 + 3, 16 + 2, 16 + 1] += eps
        ok, mx = bl.compareBrick((N,) * 3, (8,) * 3, (8,) * 3, a2, grid_dev, bIn, return_max=True)
        assert ok == want and abs(mx - eps) < 1e-9


def test_link_copy(dev):
    src = torch.arange(0, 1 << 20, dtype=torch.float64, device=dev)
    dst = torch.zeros_like(src)
    esz = 8
    spans = [(0, 512, 512), (4096, 0, 512 * 3), (70000 * 8 // 8 // 512 * 512, 8192, 512 * 130), (512 * 600, 512 * 900, 512 * 700)]
    links = [(src.data_ptr() + s * esz, dst.data_ptr() + d * esz, n * esz) for s, d, n in spans]
    LinkTable(links, dev).run()
    want = torch.zeros_like(src)
    for s, d, n in spans:
        want[d:d + n] = src[s:s + n]
    assert torch.equal(dst, want)


@pytest.mark.parametrize("name", ["mpi7pt", "mpi13pt", "mpi25pt", "mpi125pt"])
@pytest.mark.parametrize("nranks", [1, 2, 8])
def test_weak_cycle_virtual_ranks(oracle, dev, name, nranks):
    """weak/main.cpp brick_func with `nranks` ranks of a periodic cart grid living on ONE GPU (device links as the
    transport).  Oracle: the same global periodic field swept ST_ITER times as one array."""
    _weak_virtual(oracle, dev, name, nranks, (16, 16, 16), "lists" if nranks != 2 else "ranges")


@pytest.mark.parametrize("order,name,nranks,schedule", [("normal", "mpi7pt", 2, "lists"), ("bad", "mpi13pt", 8, "ranges"), ("bad", "mpi7pt_x2", 4, "deep")])
def test_weak_cycle_other_section_orders(oracle, dev, order, name, nranks, schedule):
    """the weak cycle on decompositions laid out in the reference's other section orders (skin3d_normal: 62 runs, skin3d_bad: 98
    runs, src/brick-mpi.cpp:54-136): brick numbering, run tables, sweep lists and exchange all follow the order given to initialize"""
    _weak_virtual(oracle, dev, name, nranks, (40, 32, 48) if schedule == "deep" else (16, 16, 16), schedule, order=order)


@pytest.mark.parametrize("schedule,nranks,dom", [("lists", 1, (16, 16, 16)), ("lists", 8, (24, 40, 16)), ("deep", 2, (40, 32, 48))])
def test_weak_cycle_kpaired_lists_25pt(oracle, dev, schedule, nranks, dom):
    """the weak cycle of the 25-point star over sweep lists in the k-paired order (SweepLists / TrapezoidLists with kpair=True: two
    bricks per wave, k_star_kp) -- inner box, shell slabs and walls, shrinking boxes of the deep schedule, odd level counts"""
    _weak_virtual(oracle, dev, "mpi25pt", nranks, dom, schedule, kpair=True)


@pytest.mark.parametrize("schedule,nranks", [("lists", 1), ("lists", 8), ("ranges", 2), ("deep", 4)])
def test_weak_cycle_fused_double_sweeps(oracle, dev, schedule, nranks):
    """the weak cycle with BRK_MPI7_X2 (4 fused passes = the 8 sweeps of mpi7pt between two exchanges) against the global
    periodic field swept 8 times per cycle on the CPU, every schedule"""
    _weak_virtual(oracle, dev, "mpi7pt_x2", nranks, (40, 32, 48) if schedule == "deep" else (16, 16, 16), schedule)


@pytest.mark.parametrize("name,nranks", [("mpi7pt", 2), ("mpi7pt", 8), ("mpi13pt", 4), ("mpi25pt", 1)])
def test_weak_cycle_deep_schedule_virtual_ranks(oracle, dev, name, nranks):
    """the deep-overlap schedule (drivers.weak_cycle_deep: early shrinking-box passes of every sweep, then the shells)
    against the global periodic field, on subdomains large enough that the early passes are not empty"""
    _weak_virtual(oracle, dev, name, nranks, (40, 32, 48), "deep")


@pytest.mark.parametrize("name,st", [("mpi7pt", 7), ("mpi13pt", 13), ("mpi25pt", 25), ("mpi125pt", 125), ("mpi7pt_x2", 7)])
@pytest.mark.parametrize("nranks", [1, 2, 8])
def test_weak_cycle_vs_reference_run_under_mpi(oracle, golden, dev, name, st, nranks):
    """the HIP weak cycle (exchange + ST_ITER sweeps, `nranks` virtual ranks on this GPU) against what the REFERENCE's weak driver
    produced for the same global field under mpiexec -np {1,2,8} (tests/golden/weak_cycle_ref.npz: exchangeArr + ST_CPU sweeps,
    oracle/ref_mpi_weak.cpp); the fused double-sweep cycle against the same 7-point fixture."""
    z = golden("weak_cycle_ref")
    _weak_virtual(oracle, dev, name, nranks, (16, 16, 16), "lists", seed=int(z["seed"][0]), ncycles=1,
                  ref={r: z[f"out_{st}_{nranks}_{r}"] for r in range(nranks)})


def _weak_virtual(oracle, dev, name, nranks, dom, schedule, seed=11, ncycles=2, ref=None, order="good", kpair=False):
    cart = bl.dims_create(nranks)
    gdims = [cart[2 - a] * dom[a] for a in range(3)]
    d = bl.BrickDecomp(dom, order=order)
    info = d.getBrickInfo(dev)
    gdev = grid_to_device(d.grid.reshape(-1), dev)
    ext = tuple(x + 16 for x in dom)
    stA, stB, maps, origins = {}, {}, {}, {}
    for r in range(nranks):
        coo = rank_to_coo(r, cart)
        maps[r] = dict(bl.populate(d, cart, coo))
        origins[r] = rank_origin(coo, cart, dom)
        arr = oracle.fill_array_global(ext, [o - 8 for o in origins[r]], gdims, seed)
        arr[:8], arr[-8:], arr[:, :8], arr[:, -8:], arr[:, :, :8], arr[:, :, -8:] = (-1e3,) * 6   # ghosts: junk until exchanged
        stA[r], stB[r] = info.allocate(512, dev), info.allocate(512, dev)
        bl.copyToBrick(ext, (0,) * 3, (0,) * 3, torch.from_numpy(arr).to(dev), gdev, bl.Brick(info, stA[r], 0))
    links, remote = local_links(d, stA, maps, set(range(nranks)))
    assert not remote and len(links) == {"good": 42, "normal": 62, "bad": 98}[order] * nranks
    table = LinkTable(links, dev)
    S = bl.stencil_iters(name)
    from bricklib_amd.drivers import SweepLists, TrapezoidLists, weak_cycle_deep
    lists = {"lists": lambda: SweepLists(d, dev, kpair=kpair), "ranges": lambda: None,
             "deep": lambda: TrapezoidLists(d, S, dev, kpair=kpair)}[schedule]()
    if schedule == "deep":
        assert lists.inner[0] is not None
    for _ in range(ncycles):
        table.run()                     # every rank's exchange
        for r in range(nranks):
            A, B = bl.Brick(info, stA[r], 0), bl.Brick(info, stB[r], 0)
            if schedule == "deep":
                weak_cycle_deep(name, d, A, B, lambda s: None, lists)
            else:
                weak_cycle(name, d, A, B, lambda s: None, lists=lists)
    torch.cuda.synchronize()
    # oracle: global periodic field, wrap-padded by 8 cells, S sweeps per cycle
    gext = tuple(g + 16 for g in gdims)
    cur = oracle.fill_array_global(gext, (-8, -8, -8), gdims, seed)
    oname = name[:-3] if name.endswith("_x2") else name     # a fused pass = two sweeps of the base stencil
    osweeps = S * 2 if name.endswith("_x2") else S
    R = oracle.radius(NAMES[oname])
    for _ in range(ncycles):
        for s in range(1, osweeps + 1):
            nxt = np.zeros_like(cur)
            m = s * R
            oracle.array_stencil(NAMES[oname], cur, nxt, (m,) * 3, tuple(g - m for g in gext))
            cur = nxt
        core = cur[8:-8, 8:-8, 8:-8].copy()
        cur = np.pad(core, 8, mode="wrap")
    for r in range(nranks):
        o = origins[r]
        want = core[o[2]:o[2] + dom[2], o[1]:o[1] + dom[1], o[0]:o[0] + dom[0]]
        got = torch.zeros(tuple(x + 16 for x in dom[::-1]), dtype=torch.float64, device=dev)
        bl.copyFromBrick(dom, (0,) * 3, (8,) * 3, got, gdev, bl.Brick(info, stA[r], 0))
        err = np.abs(got[8:-8, 8:-8, 8:-8].cpu().numpy() - want).max()
        assert err <= TOL, (r, err)
        if ref is not None:              # the reference's own result for this rank (run under MPI)
            err = np.abs(got[8:-8, 8:-8, 8:-8].cpu().numpy() - ref[r]).max()
            assert err <= TOL, ("vs reference run", r, err)


@pytest.mark.parametrize("variant", [1, 2, 5, 6])
@pytest.mark.parametrize("name", ["7pt", "mpi7pt"])
def test_star7_kernel_variants(oracle, dev, name, variant):
    """every 7-point kernel variant (generic / one brick per wave / 8- and 4-wave groups sharing i-faces), list and range forms,
    list lengths around the pipeline's corner cases (fewer entries than waves, odd/even trip counts)."""
    from bricklib_amd._lib import lib
    sid = bl.STENCILS[name]
    coeff = oracle.coeffs(129, 1) if name == "7pt" else None
    try:
        _set_variant_or_skip(lib, sid, variant)
        # list form on the single/ layout
        grid, grid_dev, bInfo, arr, st, bIn, ids = single_setup(oracle, dev, 48, 1024)
        hin = np.ascontiguousarray(st.dat.cpu().numpy().reshape(-1, 1024)[:, :512]).reshape(-1)
        from bricklib_amd.order import LIST_SKIP, tiled
        g3 = grid.reshape(8, 8, 8)[1:-1, 1:-1, 1:-1]
        for n in (ids.size, 1, 7, 33, 100, -1):
            st.dat.view(-1, 1024)[:, 512:] = 0
            # n < 0: tiled order of the 6x6x6 interior: rows of 6, each followed by a BRK_LIST_SKIP entry
            sub = ids[:n] if n > 0 else tiled(g3)
            if n < 0:
                assert sorted(sub[sub != LIST_SKIP].tolist()) == sorted(ids.tolist()) and (sub == LIST_SKIP).sum() >= 36
            bl.apply_stencil(name, bIn, bl.Brick(bInfo, st, 512), ids=grid_to_device(sub, dev), coeff=coeff)
            want = np.zeros_like(hin)
            oracle.brick_stencil(NAMES[name], bInfo.adj_host, hin, want, 512, ids=sub[sub != LIST_SKIP], coeff=coeff)
            got = st.dat.cpu().numpy().reshape(-1, 1024)[:, 512:]
            assert np.abs(got - want.reshape(-1, 512)).max() <= TOL, (n,)
        # range form on the BrickDecomp layout (ghost ring included: neighbours that are the dummy brick 0)
        dom = (160, 152, 168)
        d = bl.BrickDecomp(dom)
        info = d.getBrickInfo(dev)
        A, B = info.allocate(512, dev), info.allocate(512, dev)
        A.dat.uniform_(0.0, 1.0)
        A.dat[:512] = 0
        for rng_ in ((1, d.sep_pos[2]), (d.sep_pos[0], d.sep_pos[1]), (5, 6)):
            B.dat.zero_()
            bl.apply_stencil(name, bl.Brick(info, A, 0), bl.Brick(info, B, 0), brick_range=rng_, coeff=coeff)
            hin2 = A.dat.cpu().numpy()
            want = np.zeros_like(hin2)
            oracle.brick_stencil(NAMES[name], d.adj, hin2, want, 512, n=rng_[1] - rng_[0], first=rng_[0], coeff=coeff)
            assert np.abs(B.dat.cpu().numpy() - want).max() <= TOL, rng_
    finally:
        if hasattr(lib, "brk_set_variant"):
            lib.brk_set_variant(sid, 0)


@pytest.mark.parametrize("size", [1, 2, 3, 8])
@pytest.mark.parametrize("name,ghosts", [("mpi13pt", "aliased"), ("mpi13pt", "copied"), ("mpi7pt", "aliased"), ("mpi7pt", "copied")])
def test_strong_cycle_virtual_ranks(oracle, dev, name, ghosts, size):
    """strong/main.cpp shape: 64^3 global domain, 16^3 subdomains in Z-Morton order over `size` ranks (all on one GPU;
    per-peer pack -> loopback -> unpack as transport).  Oracle: the global periodic field swept on the CPU.  ghosts: runs between
    two subdomains of one rank reached through the adjacency table (the default) or copied by device links as in the reference."""
    _strong_virtual(oracle, dev, name, size, 64, 16, 21, 2, alias_local=ghosts == "aliased")


@pytest.mark.parametrize("name,size,dom,sdom", [("mpi13pt", 1, 128, 32), ("mpi13pt", 3, 128, 32), ("mpi25pt", 5, 64, 16), ("mpi125pt", 2, 64, 32)])
def test_strong_aliased_local_ghosts_bit_identical_to_copied(dev, name, size, dom, sdom):
    """two strong cycles with the same-rank ghost runs aliased through the adjacency table against the same cycles with those runs
    copied and swept (the reference's scheme): the subdomain bricks agree BIT for bit on every rank, fewer bricks are swept, and
    with one rank there is nothing left to exchange.  Both through the same kernel (plain tiled lists, kpair=False); a third run with
    the default lists -- k-paired where most groups are complete, i.e. the two-bricks-per-wave march kernel for the stars -- must
    agree with them to rounding (the march kernel contracts its sums differently: 1 ulp observed)."""
    from bricklib_amd.strong import StrongRank, loopback_transport
    res = {}
    for alias, kpair in ((True, False), (False, False), (True, True)):
        ranks = [StrongRank(dom, sdom, r, size, dev, alias_local=alias, kpair=kpair) for r in range(size)]
        gen = torch.Generator(device=dev).manual_seed(17)
        for sr in ranks:
            sr.storages[0].dat.copy_(torch.rand(sr.storages[0].dat.shape, dtype=torch.float64, device=dev, generator=gen))
        transport = loopback_transport(ranks)
        for _ in range(2):
            for sr in ranks:
                sr.exchange_begin()
            for sr in ranks:
                if sr.peers:
                    transport(sr)
            for sr in ranks:
                sr.exchange_end()
                sr.sweeps(name)
        torch.cuda.synchronize()
        res[alias, kpair] = ([sr.storages[0].dat.view(sr.nsub, sr.nb, 512)[:, 1:sr.d.sep_pos[1]].clone() for sr in ranks],
                             sum(sr.swept_bricks for sr in ranks), sum(sr.bytes_local for sr in ranks), any(sr.swept_kpaired for sr in ranks))
    for a, b in zip(res[True, False][0], res[False, False][0]):
        assert torch.equal(a, b)
    assert not res[True, False][3] and not res[False, False][3]
    for a, b in zip(res[True, False][0], res[True, True][0]):
        assert float((a - b).abs().max()) <= 4e-15
    if name == "mpi13pt" and size == 1:
        assert res[True, True][3]          # 128^3 in 32^3 on one rank: every subdomain is a 4^3 box of complete groups
    res = {True: res[True, False], False: res[False, False]}
    assert res[True][1] < res[False][1] and res[True][2] < res[False][2]      # bytes_local: links + pack + unpack copies
    if size == 1:
        assert res[True][1] == (dom // 8) ** 3 and res[True][2] == 0


@pytest.mark.parametrize("name,st,size,dom", [("mpi13pt", 13, 1, 32), ("mpi13pt", 13, 3, 32), ("mpi13pt", 13, 8, 32), ("mpi13pt", 13, 5, 64),
                                              ("mpi7pt", 7, 3, 32), ("mpi7pt", 7, 5, 64)])
def test_strong_cycle_vs_reference_run_under_mpi(oracle, golden, dev, name, st, size, dom):
    """the HIP strong cycle (Z-Morton ownership, per-peer pack -> transport -> unpack between `size` virtual ranks, ST_ITER sweeps
    over all subdomains of a rank in one launch) against what the REFERENCE's strong driver produced for the same global field
    under mpiexec (tests/golden/strong_cycle_ref.npz: its -v validation path, oracle/ref_mpi_strong.cpp)."""
    z = golden("strong_cycle_ref")
    ref = {int(k.split("_")[-1]): z[k] for k in z.files if k.startswith(f"out_{st}_{size}_{dom}_16_")}
    assert ref
    _strong_virtual(oracle, dev, name, size, dom, 16, int(z["seed"][0]), 1, ref=ref)


def _strong_virtual(oracle, dev, name, size, dom, sdom, seed, ncycles, ref=None, alias_local=True):
    from bricklib_amd.strong import StrongRank, loopback_transport
    ranks = [StrongRank(dom, sdom, r, size, dev, alias_local=alias_local) for r in range(size)]
    assert sum(sr.nsub for sr in ranks) == (dom // sdom) ** 3
    ext = (sdom + 16,) * 3
    for sr in ranks:
        for m in range(sr.nsub):
            o = sr.origin(m)
            arr = oracle.fill_array_global(ext, [x - 8 for x in o], (dom,) * 3, seed)
            arr[:8], arr[-8:], arr[:, :8], arr[:, -8:], arr[:, :, :8], arr[:, :, -8:] = (-1e3,) * 6
            bl.copyToBrick(ext, (0,) * 3, (0,) * 3, torch.from_numpy(arr).to(dev), sr.grid_dev, sr.sub_brick(0, m))
    transport = loopback_transport(ranks)
    S = bl.stencil_iters(name)
    for _ in range(ncycles):
        for sr in ranks:
            sr.exchange_begin()
        for sr in ranks:
            if sr.peers:
                transport(sr)
        for sr in ranks:
            sr.exchange_end()
            sr.sweeps(name)
    torch.cuda.synchronize()
    if size > 1:
        assert any(sr.bytes_remote > 0 for sr in ranks)
    gext = (dom + 16,) * 3
    cur = oracle.fill_array_global(gext, (-8,) * 3, (dom,) * 3, seed)
    R = oracle.radius(NAMES[name])
    for _ in range(ncycles):
        for s_ in range(1, S + 1):
            nxt = np.zeros_like(cur)
            m_ = s_ * R
            oracle.array_stencil(NAMES[name], cur, nxt, (m_,) * 3, tuple(g - m_ for g in gext))
            cur = nxt
        core = cur[8:-8, 8:-8, 8:-8].copy()
        cur = np.pad(core, 8, mode="wrap")
    for sr in ranks:
        for m in range(sr.nsub):
            o = sr.origin(m)
            want = core[o[2]:o[2] + sdom, o[1]:o[1] + sdom, o[0]:o[0] + sdom]
            got = torch.zeros(ext[::-1], dtype=torch.float64, device=dev)
            bl.copyFromBrick((sdom,) * 3, (0,) * 3, (8,) * 3, got, sr.grid_dev, sr.sub_brick(0, m))
            err = np.abs(got[8:-8, 8:-8, 8:-8].cpu().numpy() - want).max()
            assert err <= TOL, (sr.rank, m, err)
            if ref is not None and sr.l + m in ref:      # the reference's own result for this subdomain (run under MPI)
                err = np.abs(got[8:-8, 8:-8, 8:-8].cpu().numpy() - ref[sr.l + m]).max()
                assert err <= TOL, ("vs reference run", sr.rank, sr.l + m, err)


@pytest.mark.parametrize("name", list(NAMES))
def test_padded_tiled_lists_all_stencils(oracle, dev, name):
    """tiled order with BRK_LIST_SKIP padding (rows of 5 bricks) through every default kernel."""
    from bricklib_amd.order import LIST_SKIP, tiled
    grid, info = bl.init_grid((7, 9, 6), dev)
    st = bl.BrickStorage.allocate(info.nbricks, 512, dev)
    st.dat.uniform_(0.0, 1.0)
    so = bl.BrickStorage.allocate(info.nbricks, 512, dev)
    lst = tiled(grid.reshape(6, 9, 7)[1:-1, 1:-1, 1:-1])
    assert (lst == LIST_SKIP).any()
    coeff = oracle.coeffs(129, 1) if name in ("7pt", "27pt") else None
    bl.apply_stencil(name, bl.Brick(info, st, 0), bl.Brick(info, so, 0), ids=grid_to_device(lst, dev), coeff=coeff)
    hin = st.dat.cpu().numpy()
    want = np.zeros_like(hin)
    oracle.brick_stencil(NAMES[name], info.adj_host, hin, want, 512, ids=lst[lst != LIST_SKIP], coeff=coeff)
    assert np.abs(so.dat.cpu().numpy() - want).max() <= TOL


# sum of the weights of every stencil with fixed coefficients (stencils/fake.h:11-33; 125-pt: class weight x multiplicity)
_C125 = (0.1, 0.04, 0.03, 0.01, 0.006, 0.004, 0.005, 0.002, 0.003, 0.001)
_M125 = (1, 6, 6, 12, 24, 12, 8, 24, 24, 8)     # points per class sorted(|di|,|dj|,|dk|) = 000 001 002 011 012 022 111 112 122 222
WSUM = {"mpi7pt": 0.4 + 6 * 0.1, "mpi13pt": 0.4 + 6 * 0.07 + 6 * 0.03, "mpi25pt": 0.1 + 6 * (0.06 + 0.045 + 0.03 + 0.015),
        "mpi125pt": sum(c * m for c, m in zip(_C125, _M125)), "mpi7pt_x2": (0.4 + 6 * 0.1) ** 2}


@pytest.mark.parametrize("name", ["7pt", "mpi7pt", "mpi13pt", "mpi25pt", "27pt", "mpi125pt", "mpi7pt_x2"])
@pytest.mark.parametrize("form", ["list", "box"])
def test_full_size_512_properties(oracle, dev, name, form):
    """BASELINE size (512^3, 262144 bricks), every stencil of configs 2-4 (7-pt; 27-pt + 125-pt; the weak driver's mpi7pt and
    its fused double sweep; 13/25-pt), through the list entry point and the reference's box launch shape:
      * linearity  S(a*x + b*y) == a*S(x) + b*S(y)
      * a constant field maps to (sum of weights) * constant
      * 1024 random bricks agree with the CPU oracle (the fused pass: with two oracle sweeps)."""
    from bricklib_amd.order import tiled
    N, SB = 512, 66
    grid, info = bl.init_grid((SB, SB, SB), dev)
    g3 = grid.reshape(SB, SB, SB)
    ids = tiled(g3[1:-1, 1:-1, 1:-1])
    ids_dev = grid_to_device(ids, dev)
    coeff = oracle.coeffs(129, 1) if name in ("7pt", "27pt") else None
    x, y, sx, sy, sz = (bl.BrickStorage.allocate(info.nbricks, 512, dev) for _ in range(5))
    x.dat.uniform_(0.0, 1.0)
    y.dat.uniform_(0.0, 1.0)
    B = lambda s: bl.Brick(info, s, 0)
    NB = SB - 2
    where = dict(box=(0, (SB, SB, SB), (1, 1, 1), (NB, NB, NB))) if form == "box" else dict(ids=ids_dev)
    bl.apply_stencil(name, B(x), B(sx), coeff=coeff, **where)
    bl.apply_stencil(name, B(y), B(sy), coeff=coeff, **where)
    a, b = 0.75, -1.25
    z = bl.BrickStorage(a * x.dat + b * y.dat, info.nbricks, 512)
    bl.apply_stencil(name, B(z), B(sz), coeff=coeff, **where)
    lin = (sz.dat - (a * sx.dat + b * sy.dat)).abs().max().item()
    assert lin <= 1e-12, lin
    # constant field
    z.dat.fill_(2.5)
    bl.apply_stencil(name, B(z), B(sz), coeff=coeff, **where)
    if name == "7pt":
        wsum = coeff[:7].sum()
    elif name == "27pt":
        wsum = coeff[:27].sum()      # the quirk moves coeff[16] to another tap, the sum is unchanged
    else:
        wsum = WSUM[name]
    real = ids[ids != 0xFFFFFFFF]
    inner = sz.dat.view(-1, 512)[torch.from_numpy(real.astype(np.int64)).to(dev)]
    assert (inner - 2.5 * wsum).abs().max().item() <= 1e-12
    # bricks outside the swept box are untouched (sx started as zeros)
    mask = torch.ones(info.nbricks, dtype=torch.bool, device=dev)
    mask[torch.from_numpy(real.astype(np.int64)).to(dev)] = False
    assert (sx.dat.view(-1, 512)[mask] == 0).all()
    # sampled bricks vs oracle
    rng = np.random.default_rng(3)
    hin = x.dat.cpu().numpy()
    want = np.zeros_like(hin)
    if name.endswith("_x2"):
        # two oracle sweeps: the first over the sample's 27-brick neighbourhoods, the second over the sample itself, which
        # is drawn two bricks inside the box so that no first-sweep brick reads outside the grid
        sub = np.sort(rng.choice(np.ascontiguousarray(g3[3:-3, 3:-3, 3:-3]).reshape(-1), 256, replace=False)).astype(np.uint32)
        ring = np.unique(info.adj_host[sub].reshape(-1)).astype(np.uint32)
        mid = np.zeros_like(hin)
        oracle.brick_stencil(NAMES[name[:-3]], info.adj_host, hin, mid, 512, ids=ring, coeff=coeff)
        # outside `ring` the first sweep is unknown; the second sweep of `sub` only reads `ring`
        oracle.brick_stencil(NAMES[name[:-3]], info.adj_host, mid, want, 512, ids=sub, coeff=coeff)
    else:
        sub = np.sort(rng.choice(real, 1024, replace=False)).astype(np.uint32)
        oracle.brick_stencil(NAMES[name], info.adj_host, hin, want, 512, ids=sub, coeff=coeff)
    got = sx.dat.cpu().numpy().reshape(-1, 512)[sub]
    assert np.abs(got - want.reshape(-1, 512)[sub]).max() <= TOL


@pytest.mark.parametrize("name", ["7pt", "mpi7pt", "mpi13pt", "mpi25pt", "27pt", "mpi125pt"])
@pytest.mark.parametrize("shape", [(10, 10, 10), (7, 21, 5), (66, 13, 4)])
def test_box_form(oracle, dev, name, shape):
    """brk_stencil_apply_box: the reference launch shape (interior box of an init_grid numbering), ids derived in the
    kernel.  Row lengths that are / are not multiples of 8, more and fewer rows than XCDs."""
    ni, nj, nk = shape
    grid, info = bl.init_grid(shape, dev)
    st = bl.BrickStorage.allocate(info.nbricks, 1024, dev)
    st.dat.uniform_(0.0, 1.0)
    st.dat.view(-1, 1024)[:, 512:] = 0
    coeff = oracle.coeffs(129, 1) if name in ("7pt", "27pt") else None
    bl.apply_stencil(name, bl.Brick(info, st, 0), bl.Brick(info, st, 512), box=(0, shape, (1, 1, 1), (ni - 2, nj - 2, nk - 2)),
                     coeff=coeff)
    ids = np.ascontiguousarray(grid.reshape(nk, nj, ni)[1:-1, 1:-1, 1:-1]).reshape(-1)
    hin = np.ascontiguousarray(st.dat.cpu().numpy().reshape(-1, 1024)[:, :512]).reshape(-1)
    want = np.zeros_like(hin)
    oracle.brick_stencil(NAMES[name], info.adj_host, hin, want, 512, ids=ids, coeff=coeff)
    got = st.dat.cpu().numpy().reshape(-1, 1024)[:, 512:]
    assert np.abs(got - want.reshape(-1, 512)).max() <= TOL


@pytest.mark.parametrize("name", ["7pt", "mpi7pt", "cond", "mpi13pt", "mpi25pt", "27pt", "mpi125pt", "mpi7pt_x2"])
def test_reversed_sweeps_are_bit_identical(oracle, dev, name):
    """BRK_SWEEP_REVERSE (apply_stencil(reverse=True): every XCD walks its share of the launch back to front): box, list (tiled,
    k-paired, ragged with skipped entries) and range forms give the same bits as the forward sweep and the oracle's result; unknown
    flag bits and a k-paired box are refused."""
    import ctypes as C
    from bricklib_amd._lib import lib
    from bricklib_amd.order import tiled
    shape = (13, 21, 7)
    ni, nj, nk = shape
    grid, info = bl.init_grid(shape, dev)
    st = bl.BrickStorage.allocate(info.nbricks, 512, dev)
    st.dat.uniform_(0.0, 1.0)
    coeff = oracle.coeffs(129, 1) if name in ("7pt", "27pt", "cond") else None
    g3 = grid.reshape(nk, nj, ni)[1:-1, 1:-1, 1:-1]
    flat = np.ascontiguousarray(g3).reshape(-1)
    forms = {"box": dict(box=(0, shape, (1, 1, 1), (ni - 2, nj - 2, nk - 2))), "tiled": dict(ids=grid_to_device(tiled(g3), dev)),
             "kpaired": dict(ids=grid_to_device(tiled(g3, kpair=True), dev), kpaired=True), "ragged": dict(ids=grid_to_device(flat[1:-3:3].copy(), dev)),
             "range": dict(brick_range=(int(flat.min()), int(flat.max()) + 1))}
    bIn = bl.Brick(info, st, 0)
    for fname, kw in forms.items():
        out = {}
        for rev in (False, True):
            so = bl.BrickStorage.allocate(info.nbricks, 512, dev)
            bl.apply_stencil(name, bIn, bl.Brick(info, so, 0), coeff=coeff, reverse=rev, **kw)
            out[rev] = so.dat.clone()
        assert torch.equal(out[False], out[True]), (name, fname)
        if fname in ("box", "tiled", "kpaired") and name in NAMES:
            hin = st.dat.cpu().numpy()
            want = np.zeros_like(hin)
            oracle.brick_stencil(NAMES[name], info.adj_host, hin, want, 512, ids=flat, coeff=coeff)
            assert np.abs(out[True].cpu().numpy() - want).max() <= TOL, (name, fname)
    U3 = C.c_uint32 * 3
    so = bl.BrickStorage.allocate(info.nbricks, 512, dev)
    args = (bl.STENCILS[name], info.adj.data_ptr(), st.dat.data_ptr(), so.dat.data_ptr(), 512)
    cp = None if coeff is None else np.ascontiguousarray(coeff).ctypes.data_as(C.POINTER(C.c_double))
    assert lib.brk_stencil_apply_box_ex(*args, 0, U3(*shape), U3(1, 1, 1), U3(ni - 2, nj - 2, nk - 2), cp, 1, None) != 0        # k-paired box
    assert lib.brk_stencil_apply_list_ex(*args, forms["tiled"]["ids"].data_ptr(), 8, cp, 4, None) != 0                           # unknown bit
    assert lib.brk_stencil_apply_range_ex(*args, 1, 2, cp, 1, None) != 0                                                         # k-paired range


def test_box_form_decomp_inner(oracle, dev):
    """inner region [1,sep0) of a BrickDecomp as a box (base 1)"""
    d = bl.BrickDecomp((48, 40, 56))
    info = d.getBrickInfo(dev)
    A, B = info.allocate(512, dev), info.allocate(512, dev)
    A.dat.uniform_(0.0, 1.0)
    A.dat[:512] = 0
    nb = [x // 8 - 2 for x in (48, 40, 56)]
    bl.apply_stencil("mpi7pt", bl.Brick(info, A, 0), bl.Brick(info, B, 0), box=(1, nb, (0, 0, 0), nb))
    hin = A.dat.cpu().numpy()
    want = np.zeros_like(hin)
    oracle.brick_stencil(ST_MPI7, d.adj, hin, want, 512, n=d.sep_pos[0] - 1, first=1)
    assert np.abs(B.dat.cpu().numpy() - want).max() <= TOL


def test_weak_cycle_overlap_streams_single_rank(oracle, dev):
    """The overlapped schedule (exchange on a second stream, surface-first last sweep, event hand-over between cycles)
    gives the same field as the plain in-order schedule."""
    from bricklib_amd.decomp import Exchanger
    from bricklib_amd.drivers import SweepLists, TrapezoidLists, weak_cycle_deep
    dom = (64, 48, 56)
    d = bl.BrickDecomp(dom)
    bl.populate(d, [1, 1, 1], [0, 0, 0])
    info = d.getBrickInfo(dev)
    res = []
    for mode in ("plain", "overlap", "deep", "deep3", "deep_early", "deep3_early"):
        torch.manual_seed(3)
        stA, stB = info.allocate(512, dev), info.allocate(512, dev)
        stA.dat.uniform_(0.0, 1.0)
        A, B = bl.Brick(info, stA, 0), bl.Brick(info, stB, 0)
        ex = Exchanger(d, stA, 0)
        lists, deep = SweepLists(d, dev), TrapezoidLists(d, 8, dev)
        main, comm, state = torch.cuda.current_stream(), torch.cuda.Stream(), {}
        for _ in range(3):
            if mode == "plain":
                weak_cycle("mpi7pt", d, A, B, ex.exchange, lists=lists)
            elif mode == "deep":
                weak_cycle_deep("mpi7pt", d, A, B, ex.exchange, deep, main=main, comm=comm)
            elif mode == "deep3":
                weak_cycle_deep("mpi7pt", d, A, B, ex.exchange, deep, main=main, comm=comm, depth=3, full=lists)
            elif mode == "deep_early":       # round 5: the next exchange starts under the rest of the last late pass (state carried over)
                weak_cycle_deep("mpi7pt", d, A, B, ex.exchange, deep, main=main, comm=comm, state=state)
            elif mode == "deep3_early":
                weak_cycle_deep("mpi7pt", d, A, B, ex.exchange, deep, main=main, comm=comm, depth=3, full=lists, state=state)
            else:
                weak_cycle("mpi7pt", d, A, B, ex.exchange, main=main, comm=comm, overlap=True, lists=lists, state=state)
        torch.cuda.synchronize()
        res.append(stA.dat[512 * 1:512 * d.sep_pos[1]].clone())
    assert all(torch.equal(res[0], r) for r in res[1:])


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_cond_stencil(oracle, golden, dev, variant):
    """d3cond (single/hip.cpp calls d3condhip): golden fixture from the reference + oracle on a larger case, list and box forms"""
    from bricklib_amd._lib import lib
    from oracle.oracle import ST_COND7
    try:
        _set_variant_or_skip(lib, 6, variant)
        z = golden("cond_N16")
        grid, grid_dev, bInfo, arr, st, bIn, ids = single_setup(oracle, dev, 16, 512)
        st.dat -= 0.5
        so = bl.BrickStorage.allocate(bInfo.nbricks, 512, dev)
        coeff = oracle.coeffs(129, 1)
        bl.apply_stencil("cond", bIn, bl.Brick(bInfo, so, 0), ids=grid_to_device(ids, dev), coeff=coeff)
        assert np.abs(so.dat.cpu().numpy().reshape(-1, 512)[ids] - z["out"]).max() <= TOL
        so.dat.zero_()
        bl.apply_stencil("cond", bIn, bl.Brick(bInfo, so, 0), box=(0, (4, 4, 4), (1, 1, 1), (2, 2, 2)), coeff=coeff)
        assert np.abs(so.dat.cpu().numpy().reshape(-1, 512)[ids] - z["out"]).max() <= TOL
        hin = st.dat.cpu().numpy()
        want = np.zeros_like(hin)
        oracle.brick_stencil(ST_COND7, bInfo.adj_host, hin, want, 512, ids=ids, coeff=coeff)
        assert np.abs(so.dat.cpu().numpy() - want).max() <= TOL
    finally:
        if hasattr(lib, "brk_set_variant"):
            lib.brk_set_variant(6, 0)


@pytest.mark.parametrize("name,variant", [("27pt", 0), ("27pt", 1), ("27pt", 2), ("27pt", 5), ("mpi125pt", 0), ("mpi125pt", 1),
                                          ("mpi125pt", 3), ("mpi125pt", 4), ("mpi125pt", 5), ("mpi125pt", 6), ("mpi125pt", 7),
                                          ("mpi125pt", 9), ("mpi125pt", 10), ("mpi125pt", 11), ("mpi125pt", 12),
                                          ("mpi13pt", 0), ("mpi13pt", 1), ("mpi13pt", 2), ("mpi25pt", 0), ("mpi25pt", 1),
                                          ("mpi25pt", 2), ("mpi25pt", 3)])
def test_cube_kernel_variants(oracle, dev, name, variant):
    """every 13/25/27/125-point kernel (generic, padded tiles with shared i-halos at every workgroup size, dense
    conflict-free tiles with fallback slots and the two-phase path): row-major lists whose rows are not multiples of the workgroup (two runs inside one group),
    padded tiled lists, shuffled lists (a break at every entry), short lists, and BrickDecomp ranges incl. the ghost ring."""
    from bricklib_amd._lib import lib
    from bricklib_amd.order import LIST_SKIP, tiled
    sid = bl.STENCILS[name]
    coeff = oracle.coeffs(129, 1) if name == "27pt" else None
    try:
        _set_variant_or_skip(lib, sid, variant)
        grid, info = bl.init_grid((8, 7, 6), dev)
        st = bl.BrickStorage.allocate(info.nbricks, 512, dev)
        st.dat.uniform_(0.0, 1.0)
        hin = st.dat.cpu().numpy()
        g3 = grid.reshape(6, 7, 8)
        interior = np.ascontiguousarray(g3[1:-1, 1:-1, 1:-1]).reshape(-1).astype(np.uint32)
        rng = np.random.default_rng(5)
        lists = {"rows": interior, "tiled": tiled(g3[1:-1, 1:-1, 1:-1]), "shuffled": rng.permutation(interior),
                 "all-incl-boundary": np.arange(info.nbricks, dtype=np.uint32)}
        for q in (1, 3, 7, 9):
            lists["first%d" % q] = interior[:q]
        for label, lst in lists.items():
            so = bl.BrickStorage.allocate(info.nbricks, 512, dev)
            bl.apply_stencil(name, bl.Brick(info, st, 0), bl.Brick(info, so, 0), ids=grid_to_device(lst, dev), coeff=coeff)
            want = np.zeros_like(hin)
            oracle.brick_stencil(NAMES[name], info.adj_host, hin, want, 512, ids=lst[lst != LIST_SKIP], coeff=coeff)
            err = np.abs(so.dat.cpu().numpy() - want).max()
            assert err <= TOL, (label, err)
        d = bl.BrickDecomp((40, 24, 32))
        dinfo = d.getBrickInfo(dev)
        A, B = dinfo.allocate(512, dev), dinfo.allocate(512, dev)
        A.dat.uniform_(0.0, 1.0)
        A.dat[:512] = 0
        hin2 = A.dat.cpu().numpy()
        for rng_ in ((1, d.sep_pos[2]), (d.sep_pos[0], d.sep_pos[1]), (3, 4)):
            B.dat.zero_()
            bl.apply_stencil(name, bl.Brick(dinfo, A, 0), bl.Brick(dinfo, B, 0), brick_range=rng_, coeff=coeff)
            want = np.zeros_like(hin2)
            oracle.brick_stencil(NAMES[name], d.adj, hin2, want, 512, n=rng_[1] - rng_[0], first=rng_[0], coeff=coeff)
            assert np.abs(B.dat.cpu().numpy() - want).max() <= TOL, rng_
    finally:
        if hasattr(lib, "brk_set_variant"):
            lib.brk_set_variant(sid, 0)


@pytest.mark.parametrize("nranks", [2, 8])
def test_packed_exchange_tables_virtual_ranks(oracle, dev, nranks):
    """Exchanger(mode="packed") on the GPU: per-peer pack -> (loopback instead of RCCL) -> unpack gives exactly the ghost
    bricks the oracle's run-by-run exchange gives; self-neighbour runs go through the same launch as device links."""
    from bricklib_amd.decomp import Exchanger
    dom = (32, 16, 24)
    cart = bl.dims_create(nranks)
    ds, sts, exs = [], [], []
    for r in range(nranks):
        d = bl.BrickDecomp(dom)
        bl.populate(d, cart, rank_to_coo(r, cart))
        st = bl.BrickStorage((torch.arange(d.nbricks * 512, dtype=torch.float64) + r * 1e9).to(dev), d.nbricks, 512)
        ds.append(d)
        sts.append(st)
        exs.append(Exchanger(d, st, r, mode="packed"))
    assert all(len(e.recv_buf) <= 7 and e.remote for e in exs)
    for e in exs:
        e.exchange_begin()
    for r, e in enumerate(exs):           # loopback transport: my send buffer for p is p's receive buffer from me
        for p_, buf in e.send_buf.items():
            assert exs[p_].recv_buf[r].numel() == buf.numel()
            exs[p_].recv_buf[r].copy_(buf)
    for e in exs:
        e.exchange_end()
    torch.cuda.synchronize()
    od = oracle.decomp(dom)
    want = [np.arange(od.nbricks * 512, dtype=np.float64) + r * 1e9 for r in range(nranks)]
    oracle.exchange(od, want, [oracle.populate_rank_map(cart, rank_to_coo(r, cart)) for r in range(nranks)], 512)
    for r in range(nranks):
        assert (sts[r].dat.cpu().numpy() == want[r]).all(), r


@pytest.mark.parametrize("nranks,mode", [(2, "packed"), (8, "packed"), (4, "direct"), (8, "direct"), (3, "shift"), (8, "shift")])
def test_c_abi_exchange_views_loopback(oracle, dev, nranks, mode):
    """The C-ABI exchange (brk_xplan_build_decomp -> brk_exchange_view_create -> begin / transport / end) for every rank
    of a periodic grid living on this one GPU, with brk_exchange_loopback standing in for the RCCL group: storages equal
    the oracle's run-by-run exchange bit for bit, in all three modes."""
    from bricklib_amd.decomp import Exchanger, exchange_loopback
    dom = (32, 16, 24)
    cart = bl.dims_create(nranks)
    sts, exs = [], []
    for r in range(nranks):
        d = bl.BrickDecomp(dom)
        bl.populate(d, cart, rank_to_coo(r, cart))
        st = bl.BrickStorage((torch.arange(d.nbricks * 512, dtype=torch.float64) + r * 1e9).to(dev), d.nbricks, 512)
        sts.append(st)
        exs.append(Exchanger(d, st, r, mode=mode))
    assert any(e.messages for e in exs)
    if mode == "packed":
        assert all(e.messages <= 7 for e in exs)
    if mode == "shift":
        assert all(e.nstages == 3 and e.messages <= 6 for e in exs)
    for stage in range(exs[0].nstages):
        for e in exs:
            e.exchange_begin(stage=stage)
        exchange_loopback(exs, stage)
        for e in exs:
            e.exchange_end(stage=stage)
    torch.cuda.synchronize()
    od = oracle.decomp(dom)
    want = [np.arange(od.nbricks * 512, dtype=np.float64) + r * 1e9 for r in range(nranks)]
    oracle.exchange(od, want, [oracle.populate_rank_map(cart, rank_to_coo(r, cart)) for r in range(nranks)], 512)
    for r in range(nranks):
        assert (sts[r].dat.cpu().numpy() == want[r]).all(), r


@pytest.mark.timeout(240)
@pytest.mark.parametrize("mode", ["packed", "direct", "shift"])
def test_c_abi_exchange_rccl_single_rank(oracle, dev, mode):
    """The RCCL transport itself on one GPU: a 1-rank communicator created through the C ABI (brk_comm_unique_id +
    brk_comm_init_rank), a periodic 1x1x1 decomposition whose self-neighbour runs are forced through the communicator
    (BRK_XCHG_SELF_VIA_COMM: pack -> ncclGroupStart; ncclRecv/ncclSend to self; ncclGroupEnd -> unpack, all inside
    libbrk.so) -- equals the oracle's exchange bit for bit.  Also the reference-shaped call brk_exchange(decomp, storage,
    step, comm, rank_of_neighbor, stream)."""
    from bricklib_amd.decomp import Comm, Exchanger, allneighbors
    from bricklib_amd._lib import check, lib
    dom = (40, 24, 32)
    d = bl.BrickDecomp(dom)
    bl.populate(d, [1, 1, 1], [0, 0, 0])
    base = torch.arange(d.nbricks * 512, dtype=torch.float64)
    od = oracle.decomp(dom)
    want = [base.numpy().copy()]
    oracle.exchange(od, want, [oracle.populate_rank_map([1, 1, 1], [0, 0, 0])], 512)
    comm = Comm.init_rank(Comm.unique_id(), 0, 1)
    try:
        st = bl.BrickStorage(base.clone().to(dev), d.nbricks, 512)
        ex = Exchanger(d, st, 0, mode=mode, comm=comm, self_via_comm=True)
        assert ex.messages > 0
        if mode != "shift":
            assert ex.bytes_remote == sum(g["len"] for g in d.ghost) * 4096       # every run travels through the communicator
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        ex.exchange(side)                          # on a second stream, as the drivers do
        side.synchronize()
        assert (st.dat.cpu().numpy() == want[0]).all()
        if mode == "packed":
            st2 = bl.BrickStorage(base.clone().to(dev), d.nbricks, 512)
            ranks = (C.c_int * 27)(*[d.rank_map[n] for n in allneighbors()])
            check(lib.brk_exchange(d._h, C.c_void_p(st2.dat.data_ptr()), 512, comm._h, ranks, None))
            torch.cuda.synchronize()
            assert (st2.dat.cpu().numpy() == want[0]).all()
            st3 = bl.BrickStorage(base.clone().to(dev), d.nbricks, 512)       # no communicator: every neighbour is this rank
            check(lib.brk_exchange(d._h, C.c_void_p(st3.dat.data_ptr()), 512, None, ranks, None))
            torch.cuda.synchronize()
            assert (st3.dat.cpu().numpy() == want[0]).all()
    finally:
        comm.destroy()


@pytest.mark.parametrize("name", ["7pt", "mpi13pt"])
def test_large_1024_cubed_addressing(oracle, dev, name):
    """1024^3 cells = 2.2 M bricks, 9 GB per field: byte offsets beyond 2^32 in every kernel path (id list, 64-bit brick
    base + 32-bit lane offsets).  Constant-field property over the whole grid + bricks sampled from the top of the id
    range against the oracle on a compacted copy of their neighbourhoods."""
    from bricklib_amd.order import tiled
    SB = 130
    grid, info = bl.init_grid((SB, SB, SB), dev)
    ids = tiled(grid.reshape(SB, SB, SB)[1:-1, 1:-1, 1:-1])
    ids_dev = grid_to_device(ids, dev)
    coeff = oracle.coeffs(129, 1) if name == "7pt" else None
    x, sx = (bl.BrickStorage.allocate(info.nbricks, 512, dev) for _ in range(2))
    assert x.dat.numel() * 8 > 2 ** 33
    B = lambda s: bl.Brick(info, s, 0)
    x.dat.fill_(1.5)
    bl.apply_stencil(name, B(x), B(sx), ids=ids_dev, coeff=coeff)
    wsum = coeff[:7].sum() if name == "7pt" else 0.4 + 6 * 0.07 + 6 * 0.03
    real = torch.from_numpy(ids[ids != 0xFFFFFFFF].astype(np.int64)).to(dev)
    top = real[real > info.nbricks - 3 * SB * SB]            # the last interior planes: offsets > 8.9 GB
    assert top.numel() > 0 and (sx.dat.view(-1, 512)[top] - 1.5 * wsum).abs().max().item() <= 1e-12
    assert (sx.dat.view(-1, 512)[real[:: 997]] - 1.5 * wsum).abs().max().item() <= 1e-12
    # random data, sampled bricks: compact their 27-neighbourhoods into a small storage the oracle can take
    x.dat.uniform_(0.0, 1.0)
    bl.apply_stencil(name, B(x), B(sx), ids=ids_dev, coeff=coeff)
    rng = np.random.default_rng(9)
    real_h = real.cpu().numpy()
    sub = np.concatenate([rng.choice(real_h, 192, replace=False), rng.choice(top.cpu().numpy(), 64, replace=False)])
    nb = info.adj_host[sub]                                   # [n][27] global ids
    uniq, inv = np.unique(nb.reshape(-1), return_inverse=True)
    small_adj = np.zeros((uniq.size, 27), dtype=np.uint32)   # only the sampled bricks' rows are used
    local_ids = inv.reshape(-1, 27)[:, 13].astype(np.uint32)
    small_adj[local_ids] = inv.reshape(-1, 27).astype(np.uint32)
    hin = x.dat.view(-1, 512)[torch.from_numpy(uniq.astype(np.int64)).to(dev)].cpu().numpy().reshape(-1)
    want = np.zeros_like(hin)
    oracle.brick_stencil(NAMES[name], small_adj, hin, want, 512, ids=local_ids, coeff=coeff)
    got = sx.dat.view(-1, 512)[torch.from_numpy(sub.astype(np.int64)).to(dev)].cpu().numpy()
    assert np.abs(got - want.reshape(-1, 512)[local_ids]).max() <= TOL


def _taps_cases(oracle):
    rng = np.random.default_rng(17)
    coeff = oracle.coeffs(129, 1)
    seven = [((0, 0, 0), coeff[0]), ((1, 0, 0), coeff[1]), ((-1, 0, 0), coeff[2]), ((0, 1, 0), coeff[3]), ((0, -1, 0), coeff[4]),
             ((0, 0, 1), coeff[5]), ((0, 0, -1), coeff[6])]
    cube = lambda r: [((di, dj, dk), float(rng.uniform(-1, 1))) for dk in range(-r, r + 1) for dj in range(-r, r + 1)
                      for di in range(-r, r + 1)]
    c27, c125 = cube(1), cube(2)
    lopsided = [c125[q] for q in rng.choice(125, 19, replace=False)]
    # star-shaped tables beyond the 125-point cube (round 4): a radius-3 and a radius-4 star with their own weights (NOT
    # mpi25pt's five classes), a lopsided one, and single far taps
    def star(r):
        t = [((0, 0, 0), float(rng.uniform(-1, 1)))]
        for a in range(3):
            for o in range(-r, r + 1):
                if o:
                    off = [0, 0, 0]
                    off[a] = o
                    t.append((tuple(off), float(rng.uniform(-1, 1))))
        return t
    c343, c729 = cube(3), cube(4)
    return {"seven": seven, "cube27": c27, "cube125": c125, "lopsided19": lopsided, "single-shifted": [((2, -2, 1), 1.0)],
            # cube-shaped tables beyond the 125-point cube (round 5): all 7^3 / 9^3 taps, sparse off-axis ones, a long star (> 64 taps)
            "cube343": c343, "cube729": c729, "r3-sparse": [c343[q] for q in rng.choice(343, 23, replace=False)] + [((3, -3, 3), 0.5)],
            "r4-sparse": [c729[q] for q in rng.choice(729, 17, replace=False)] + [((-4, 4, -4), -0.25)],
            "star-r4-long": star(4) * 3, "star-r2": star(2),
            "star-r3": star(3), "star-r4": star(4), "star-lopsided": [((4, 0, 0), 0.5), ((0, -3, 0), -0.25), ((0, 0, 4), 2.0), ((0, 0, -4), 1.0), ((0, 0, 0), -1.0)],
            "far-i": [((-4, 0, 0), 1.0)], "far-k": [((0, 0, 3), 1.0)]}


@pytest.mark.parametrize("case", ["seven", "cube27", "cube125", "lopsided19", "single-shifted", "star-r3", "star-r4", "star-lopsided", "far-i", "far-k",
                                  "cube343", "cube729", "r3-sparse", "r4-sparse", "star-r4-long", "star-r2"])
def test_stencil_given_as_taps(oracle, dev, case):
    """brk_stencil_apply_taps: arbitrary linear stencils up to radius 4 -- inside the 125-point cube, star-shaped, and
    (round 5) cube-shaped to 9^3 taps with the table in device memory (the reference generator takes any script,
    codegen/st/codegen/base.py:678) --, list (with skipped entries) and range forms, against the oracle's accessor-semantics
    evaluation; the 7-point taps also against the built-in kernel."""
    from bricklib_amd.order import LIST_SKIP, tiled
    from bricklib_amd.stencils import apply_taps
    taps = _taps_cases(oracle)[case]
    grid, info = bl.init_grid((9, 7, 6), dev)
    st = bl.BrickStorage.allocate(info.nbricks, 512, dev)
    st.dat.uniform_(0.0, 1.0)
    so = bl.BrickStorage.allocate(info.nbricks, 512, dev)
    lst = tiled(grid.reshape(6, 7, 9)[1:-1, 1:-1, 1:-1])
    assert (lst == LIST_SKIP).any()
    apply_taps(taps, bl.Brick(info, st, 0), bl.Brick(info, so, 0), ids=grid_to_device(lst, dev))
    hin = st.dat.cpu().numpy()
    want = np.zeros_like(hin)
    real = lst[lst != LIST_SKIP]
    oracle.brick_taps(info.adj_host, hin, want, 512, real, taps)
    tol = TOL * max(1.0, len(taps) / 125)      # hundreds of terms of either sign
    assert np.abs(so.dat.cpu().numpy() - want).max() <= tol
    if case in ("cube343", "r3-sparse", "cube729", "r4-sparse", "star-r4-long"):
        assert bl.last_kernel() == ("k_taps_far<3,2>" if case in ("cube343", "r3-sparse") else "k_taps_far<4,1>")
    if case.startswith("star-r") and case != "star-r4-long":
        assert bl.last_kernel() == ("k_star_taps<2,2>" if case == "star-r2" else "k_star_taps<4,1>")
    if case == "seven":
        sb = bl.BrickStorage.allocate(info.nbricks, 512, dev)
        bl.apply_stencil("7pt", bl.Brick(info, st, 0), bl.Brick(info, sb, 0), ids=grid_to_device(lst, dev), coeff=oracle.coeffs(129, 1))
        assert (sb.dat - so.dat).abs().max().item() <= TOL
    # range form over a BrickDecomp numbering incl. the ghost ring (neighbours that are the dummy brick 0)
    d = bl.BrickDecomp((24, 16, 32))
    dinfo = d.getBrickInfo(dev)
    A, B = dinfo.allocate(512, dev), dinfo.allocate(512, dev)
    A.dat.uniform_(0.0, 1.0)
    A.dat[:512] = 0
    apply_taps(taps, bl.Brick(dinfo, A, 0), bl.Brick(dinfo, B, 0), brick_range=(1, d.sep_pos[2]))
    hin2 = A.dat.cpu().numpy()
    want2 = np.zeros_like(hin2)
    oracle.brick_taps(d.adj, hin2, want2, 512, np.arange(1, d.sep_pos[2], dtype=np.uint32), taps)
    assert np.abs(B.dat.cpu().numpy() - want2).max() <= TOL


EXPR_HEAD = """
from st.expr import Index, ConstRef, If
from st.grid import Grid
from st.func import Func
i = Index(0); j = Index(1); k = Index(2)
"""
EXPR_SCRIPTS = {
    # the shape of the reference's stencils/cond.py, term order of its 7-point loop (3axis.cpp:104-111)
    "cond": EXPR_HEAD + """
pos = Func("max", 2)
u = Grid("in", 3); v = Grid("out", 3)
c = [ConstRef("coeff[%d]" % q) for q in range(7)]
zero = ConstRef("0.0")
total = None
for q, (di, dj, dk) in ((5, (0, 0, 1)), (6, (0, 0, -1)), (3, (0, 1, 0)), (4, (0, -1, 0)), (1, (1, 0, 0)), (2, (-1, 0, 0)), (0, (0, 0, 0))):
    term = c[q] * pos(u(i + di, j + dj, k + dk), zero)
    total = term if total is None else total + term
v(i, j, k).assign(If(total > 0, total, -total))
STENCIL = [v]
""",
    # a flux limiter: products of grid references, min / abs / comparisons, radius 2, taps off the axes
    "limiter": EXPR_HEAD + """
lo = Func("min", 2); mag = Func("fabs", 1)
u = Grid("u", 3); v = Grid("v", 3)
dl = u(i, j, k) - u(i - 1, j, k); dr = u(i + 1, j, k) - u(i, j, k)
far = u(i + 2, j - 1, k) * u(i - 2, j, k + 2) - u(i, j + 2, k - 2)
slope = If(dl * dr > 0, lo(mag(dl), mag(dr)), 0.0)
v(i, j, k).assign(u(i, j, k) + ConstRef("cfl") * slope * If(far >= 0, 1.0, -1.0))
STENCIL = [v]
""",
    # taps beyond the 125-point cube (radius 3 and 4 tiles)
    "far3": EXPR_HEAD + """
hi = Func("max", 2)
u = Grid("u", 3); v = Grid("v", 3)
v(i, j, k).assign(hi(u(i + 3, j - 3, k + 1), u(i - 2, j + 3, k - 3)) * u(i, j, k) - u(i - 3, j, k))
STENCIL = [v]
""",
    "far4": EXPR_HEAD + """
lo = Func("min", 2)
u = Grid("u", 3); v = Grid("v", 3)
v(i, j, k).assign(If(u(i + 4, j - 4, k + 4) > u(i - 4, j + 1, k - 2), lo(u(i, j + 4, k), u(i, j, k - 4)), u(i - 1, j, k) * u(i, j, k)))
STENCIL = [v]
""",
    # star-shaped beyond radius 1 (finite-difference scripts of higher order with limiters): the star kernel's cell update when compiled
    "star2": EXPR_HEAD + """
lo = Func("min", 2); mag = Func("fabs", 1)
u = Grid("u", 3); v = Grid("v", 3)
d4 = lambda a, b, c, d: (8.0 * (c - b) - (d - a)) / 12.0
gx = d4(u(i - 2, j, k), u(i - 1, j, k), u(i + 1, j, k), u(i + 2, j, k))
gy = d4(u(i, j - 2, k), u(i, j - 1, k), u(i, j + 1, k), u(i, j + 2, k))
gz = d4(u(i, j, k - 2), u(i, j, k - 1), u(i, j, k + 1), u(i, j, k + 2))
v(i, j, k).assign(u(i, j, k) - ConstRef("cfl") * (lo(mag(gx), mag(gy)) + If(gz > 0, gz, 0.5 * gz)))
STENCIL = [v]
""",
    "star4": EXPR_HEAD + """
hi = Func("max", 2)
u = Grid("u", 3); v = Grid("v", 3)
v(i, j, k).assign(hi(u(i + 4, j, k) - u(i - 3, j, k), u(i, j - 4, k) * u(i, j + 3, k)) + u(i, j, k + 4) - 0.5 * u(i, j, k - 4) + u(i, j, k))
STENCIL = [v]
""",
    # division and square root
    "norm": EXPR_HEAD + """
root = Func("sqrt", 1)
u = Grid("u", 3); v = Grid("v", 3)
gx = u(i + 1, j, k) - u(i - 1, j, k); gy = u(i, j + 1, k) - u(i, j - 1, k); gz = u(i, j, k + 1) - u(i, j, k - 1)
v(i, j, k).assign(u(i, j, k) / (ConstRef("eps") + root(gx * gx + gy * gy + gz * gz)))
STENCIL = [v]
""",
}


@pytest.mark.parametrize("case", ["cond", "limiter", "norm", "far3", "far4", "star2", "star4"])
def test_stencil_given_as_expression(oracle, dev, case):
    """brk_stencil_apply_expr (non-linear scripts: Func / If / products, what stencils/cond.py is made of): list (with
    skipped entries) and range forms against the oracle's evaluation of the same tree with the accessor's neighbour semantics.  The
    interpreter does the script's operations one by one in fp64 (no contraction across operations), so +, -, *, max, min, abs, If
    agree BIT FOR BIT; / and sqrt within 2 ulp.  The cond-shaped script also against the tuned built-in kernel."""
    from bricklib_amd.order import LIST_SKIP, tiled
    from bricklib_amd.spec import load_spec
    spec = load_spec(EXPR_SCRIPTS[case])
    coeff = oracle.coeffs(129, 1) - 0.45           # both signs: both arms of cond's If are taken
    consts = {"coeff": coeff, "cfl": 0.3, "eps": 1e-3}
    assert not spec.linear

    def agree(got, want):
        if case in ("norm", "star2"):      # division / square root: within 2 ulp of the host's
            assert np.abs(got - want).max() <= 4e-16 * max(1.0, np.abs(want).max()) and np.abs(want).max() > 0.4
        else:
            assert np.array_equal(got, want) and np.unique(want).size > 1000

    grid, info = bl.init_grid((9, 7, 6), dev)
    st = bl.BrickStorage.allocate(info.nbricks, 512, dev)
    st.dat.uniform_(-0.5, 0.5)
    so = bl.BrickStorage.allocate(info.nbricks, 512, dev)
    lst = tiled(grid.reshape(6, 7, 9)[1:-1, 1:-1, 1:-1])
    assert (lst == LIST_SKIP).any()
    bl.apply_expr(spec, bl.Brick(info, st, 0), bl.Brick(info, so, 0), ids=grid_to_device(lst, dev), constants=consts, interpret=True)
    assert bl.last_kernel() == {1: "k_expr<1,4>", 2: "k_expr<2,4>", 3: "k_expr<3,2>", 4: "k_expr<4,1>"}[spec.radius]
    hin = st.dat.cpu().numpy()
    want = np.zeros_like(hin)
    real = lst[lst != LIST_SKIP]
    oracle.brick_expr(spec, info.adj_host, hin, want, 512, real, consts)
    agree(so.dat.cpu().numpy(), want)
    # the same program compiled at run time (brk_expr_compile): same operations, one rounding each -> the interpreter's numbers
    ck = bl.CompiledExpr(spec, consts)
    sc = bl.BrickStorage.allocate(info.nbricks, 512, dev)
    ck.apply(bl.Brick(info, st, 0), bl.Brick(info, sc, 0), ids=grid_to_device(lst, dev))
    star = all(sum(1 for x in o["offset"] if x) <= 1 for o in spec.program(consts)[0])       # every tap on one axis
    # star-shaped of radius 1: the 7-point kernel k_s7_wg around the program; other stars: the star kernel k_star_wg; else the cube tile
    form = ",s7,list" if star and spec.radius == 1 else (",star_wg" if star else (",cube_dense" if spec.radius <= 2 else ""))
    assert bl.last_kernel() == "k_expr_jit<%d%s>" % (spec.radius, form) and star == (case in ("cond", "norm", "star2", "star4"))
    assert torch.equal(sc.dat, so.dat) and ("brk_user_point7" if form.startswith(",s7") else "brk_user_star" if star else "brk_user_cube" if form else "brick_apply_fn<") in ck.source
    if case == "cond":
        sb = bl.BrickStorage.allocate(info.nbricks, 512, dev)
        bl.apply_stencil("cond", bl.Brick(info, st, 0), bl.Brick(info, sb, 0), ids=grid_to_device(lst, dev), coeff=coeff)
        assert (sb.dat - so.dat).abs().max().item() <= TOL
    # range form over a BrickDecomp numbering incl. the ghost ring (neighbours that are the dummy brick 0)
    d = bl.BrickDecomp((24, 16, 32))
    dinfo = d.getBrickInfo(dev)
    A, B = dinfo.allocate(512, dev), dinfo.allocate(512, dev)
    A.dat.uniform_(-0.5, 0.5)
    A.dat[:512] = 0
    bl.apply_expr(spec, bl.Brick(dinfo, A, 0), bl.Brick(dinfo, B, 0), brick_range=(1, d.sep_pos[2]), constants=consts, interpret=True)
    hin2 = A.dat.cpu().numpy()
    want2 = np.zeros_like(hin2)
    oracle.brick_expr(spec, d.adj, hin2, want2, 512, np.arange(1, d.sep_pos[2], dtype=np.uint32), consts)
    agree(B.dat.cpu().numpy(), want2)
    C = dinfo.allocate(512, dev)
    ck.apply(bl.Brick(dinfo, A, 0), bl.Brick(dinfo, C, 0), brick_range=(1, d.sep_pos[2]))
    assert torch.equal(C.dat, B.dat) and bl.last_kernel() == "k_expr_jit<%d%s>" % (spec.radius, form.replace("list", "range"))
    ck.release()


def test_compiled_expression_lifecycle(dev):
    """brk_expr_compile: one kernel per distinct program (a second compile returns the first), other constants = another kernel,
    brk_finalize unloads the modules (a kernel used afterwards is refused, compiling the program again revives it), release twice is
    reported"""
    import ctypes as C
    from bricklib_amd._lib import BrkError, lib
    from bricklib_amd.spec import load_spec
    spec = load_spec(EXPR_SCRIPTS["limiter"])
    grid, info = bl.init_grid((4, 4, 4), dev)
    st, s1, s2 = (bl.BrickStorage.allocate(info.nbricks, 512, dev) for _ in range(3))
    st.dat.uniform_(-0.5, 0.5)
    a = bl.CompiledExpr(spec, {"cfl": 0.3})
    b = bl.CompiledExpr(spec, {"cfl": 0.3})
    c = bl.CompiledExpr(spec, {"cfl": 0.25})
    assert a._h.value == b._h.value and c._h.value != a._h.value
    a.apply(bl.Brick(info, st, 0), bl.Brick(info, s1, 0))
    c.apply(bl.Brick(info, st, 0), bl.Brick(info, s2, 0))
    assert not torch.equal(s1.dat, s2.dat)
    bl.apply_expr(spec, bl.Brick(info, st, 0), bl.Brick(info, s2, 0), constants={"cfl": 0.3}, interpret=True)
    assert torch.equal(s1.dat, s2.dat) and bl.last_kernel() == "k_expr<2,4>"
    # apply_expr's default: compiled on first use, kept on the spec per set of constants
    for cfl in (0.3, 0.3, 0.25):
        s2.dat.zero_()
        bl.apply_expr(spec, bl.Brick(info, st, 0), bl.Brick(info, s2, 0), constants={"cfl": cfl})
        assert bl.last_kernel() == "k_expr_jit<2,cube_dense>" and torch.equal(s1.dat, s2.dat) == (cfl == 0.3)
    assert len(spec._compiled) == 2 and {k._h.value for k in spec._compiled.values()} == {a._h.value, c._h.value}
    with pytest.raises(BrkError):
        a.apply(bl.Brick(info, st, 0), bl.Brick(info, st, 0))          # in == out
    bl.finalize()
    with pytest.raises(BrkError, match="unloaded"):
        a.apply(bl.Brick(info, st, 0), bl.Brick(info, s1, 0))
    again = bl.CompiledExpr(spec, {"cfl": 0.3})
    assert again._h.value == a._h.value
    s2.dat.zero_()
    a.apply(bl.Brick(info, st, 0), bl.Brick(info, s2, 0))
    assert torch.equal(s1.dat, s2.dat)
    handle = c._h.value
    for k in (a, b, c, again, *spec._compiled.values()):
        k.release()
    assert lib.brk_expr_release(C.c_void_p(handle)) != 0               # gone already


def test_scripts_at_baseline_size_512(oracle, dev):
    """BASELINE size (512^3, 262144 bricks) for stencils given as data: the cond-shaped script compiled at run time against the tuned
    d3cond kernel over the whole field; a homogeneous script (degree 1: |sum c max(u, 0)|) scales with positive factors; the compiled
    and the interpreted form agree bit for bit on every brick; 512 random bricks against the oracle's evaluation of the tree; bricks
    outside the list are untouched."""
    from bricklib_amd.order import tiled
    from bricklib_amd.spec import load_spec
    SB = 66
    grid, info = bl.init_grid((SB, SB, SB), dev)
    g3 = grid.reshape(SB, SB, SB)
    ids = tiled(g3[1:-1, 1:-1, 1:-1])
    ids_dev = grid_to_device(ids, dev)
    coeff = oracle.coeffs(129, 1) - 0.45
    spec = load_spec(EXPR_SCRIPTS["cond"])
    x, a, b, c = (bl.BrickStorage.allocate(info.nbricks, 512, dev) for _ in range(4))
    x.dat.uniform_(-0.5, 0.5)
    B = lambda s: bl.Brick(info, s, 0)
    ck = bl.CompiledExpr(spec, {"coeff": coeff})
    ck.apply(B(x), B(a), ids=ids_dev)
    bl.apply_stencil("cond", B(x), B(b), ids=ids_dev, coeff=coeff)
    assert (a.dat - b.dat).abs().max().item() <= TOL and a.dat.abs().max().item() > 0.1
    bl.apply_expr(spec, B(x), B(c), ids=ids_dev, constants={"coeff": coeff}, interpret=True)
    assert torch.equal(a.dat, c.dat)
    real = ids[ids != 0xFFFFFFFF]
    mask = torch.ones(info.nbricks, dtype=torch.bool, device=dev)
    mask[torch.from_numpy(real.astype(np.int64)).to(dev)] = False
    assert (a.dat.view(-1, 512)[mask] == 0).all()
    z = bl.BrickStorage(3.0 * x.dat, info.nbricks, 512)
    ck.apply(B(z), B(c), ids=ids_dev)
    assert (c.dat - 3.0 * a.dat).abs().max().item() <= 1e-13
    rng = np.random.default_rng(5)
    sub = np.sort(rng.choice(real, 512, replace=False)).astype(np.uint32)
    hin = x.dat.cpu().numpy()
    want = np.zeros_like(hin)
    oracle.brick_expr(spec, info.adj_host, hin, want, 512, sub, {"coeff": coeff})
    got = a.dat.view(-1, 512)[torch.from_numpy(sub.astype(np.int64)).to(dev)].cpu().numpy()
    assert np.array_equal(got, want.reshape(-1, 512)[sub])
    ck.release()


@pytest.mark.parametrize("seed", [1, 2, 3, 5, 8, 13, 21, 34])
def test_random_scripts_interpreted_compiled_and_oracle_agree(oracle, dev, seed):
    """seeded random expression DAGs (tests/random_scripts.py; radius 1..4, shared sub-expressions, If / comparisons / max / min /
    abs): brk_stencil_apply_expr, the run-time compiled kernel and the oracle's evaluation of the tree agree BIT FOR BIT on a BrickDecomp
    numbering incl. the ghost ring (neighbours that are the dummy brick 0)"""
    from random_scripts import random_script
    from bricklib_amd.spec import load_spec
    spec = load_spec(random_script(seed, radius=1 + seed % 4, steps=30))
    consts = {"cA": 0.3}
    d = bl.BrickDecomp((24, 16, 32))
    dinfo = d.getBrickInfo(dev)
    A, B, C = (dinfo.allocate(512, dev) for _ in range(3))
    A.dat.uniform_(-0.5, 0.5)
    A.dat[:512] = 0
    rng = (1, d.sep_pos[2])
    bl.apply_expr(spec, bl.Brick(dinfo, A, 0), bl.Brick(dinfo, B, 0), brick_range=rng, constants=consts, interpret=True)
    ck = bl.CompiledExpr(spec, consts)
    ck.apply(bl.Brick(dinfo, A, 0), bl.Brick(dinfo, C, 0), brick_range=rng)
    ck.release()
    assert torch.equal(B.dat, C.dat)
    hin = A.dat.cpu().numpy()
    want = np.zeros_like(hin)
    oracle.brick_expr(spec, d.adj, hin, want, 512, np.arange(rng[0], rng[1], dtype=np.uint32), consts)
    got = B.dat.cpu().numpy()
    assert np.array_equal(got, want) and np.unique(want).size > 1000


def test_scripts_over_interleaved_fields(oracle, dev):
    """the reference's single/ drivers keep bIn and bOut INTERLEAVED in one storage (step 1024, offsets 0 and 512, 3axis.cpp:26-28):
    stencils given as data -- a radius-3 cube table, an expression interpreted and compiled -- read one field and write the other of the
    same storage, and leave the input halves untouched"""
    from bricklib_amd.spec import load_spec
    from bricklib_amd.stencils import apply_taps
    shape = (9, 7, 6)
    grid, info = bl.init_grid(shape, dev)
    ids = np.ascontiguousarray(grid.reshape(6, 7, 9)[1:-1, 1:-1, 1:-1]).reshape(-1)
    ids_dev = grid_to_device(ids, dev)
    rng = np.random.default_rng(11)
    taps = [((int(a), int(b), int(c)), float(w)) for a, b, c, w in zip(*(rng.integers(-3, 4, 40) for _ in range(3)), rng.uniform(-1, 1, 40))] + [((3, -3, 2), 0.5)]
    spec = load_spec(EXPR_SCRIPTS["limiter"])
    consts = {"cfl": 0.3}
    ck = bl.CompiledExpr(spec, consts)
    runs = {"taps": lambda A, B: apply_taps(taps, A, B, ids=ids_dev), "interpreted": lambda A, B: bl.apply_expr(spec, A, B, ids=ids_dev, constants=consts, interpret=True),
            "compiled": lambda A, B: ck.apply(A, B, ids=ids_dev)}
    for name, run in runs.items():
        st = bl.BrickStorage.allocate(info.nbricks, 1024, dev)
        st.dat.uniform_(-0.5, 0.5)
        st.dat.view(-1, 1024)[:, 512:] = 0
        before = st.dat.view(-1, 1024)[:, :512].clone()
        run(bl.Brick(info, st, 0), bl.Brick(info, st, 512))
        assert torch.equal(st.dat.view(-1, 1024)[:, :512], before), name
        hin = np.ascontiguousarray(before.cpu().numpy()).reshape(-1)
        want = np.zeros_like(hin)
        if name == "taps":
            oracle.brick_taps(info.adj_host, hin, want, 512, ids, taps)
        else:
            oracle.brick_expr(spec, info.adj_host, hin, want, 512, ids, consts)
        got = st.dat.cpu().numpy().reshape(-1, 1024)[:, 512:]
        if name == "taps":
            assert np.abs(got - want.reshape(-1, 512)).max() <= TOL and np.abs(want).max() > 0.1
        else:
            assert np.array_equal(got, want.reshape(-1, 512)), name
    ck.release()


def test_compiled_kernels_are_kept_on_disk_across_processes(tmp_path):
    """BRK_JIT_CACHE_DIR: the first process compiles a script and leaves the code object in the directory, the second one loads it
    (brk_expr_kernel_from_disk) and computes the same field; a file that is not a code object is ignored and replaced"""
    import subprocess
    import sys
    prog = """
import sys, torch
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import bricklib_amd as bl
from bricklib_amd.spec import load_spec
from random_scripts import random_script
dev = torch.device("cuda:0")
spec = load_spec(random_script(4, radius=2, steps=20))
ck = bl.CompiledExpr(spec, {"cA": 0.3})
grid, info = bl.init_grid((4, 4, 4), dev)
a, b = (bl.BrickStorage.allocate(info.nbricks, 512, dev) for _ in range(2))
torch.manual_seed(3)
a.dat.copy_(torch.rand(a.dat.shape, dtype=torch.float64) - 0.5)
ck.apply(bl.Brick(info, a, 0), bl.Brick(info, b, 0))
print("from_disk", int(ck.from_disk), "checksum", repr(b.dat.sum().item()))
ck.release()
bl.finalize()
""" % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, BRK_JIT_CACHE_DIR=str(tmp_path))

    def run():
        r = subprocess.run([sys.executable, "-c", prog], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        out = r.stdout.decode()
        assert r.returncode == 0, out[-2000:]
        line = [l for l in out.splitlines() if l.startswith("from_disk")][-1].split()
        return int(line[1]), line[3]
    first = run()
    files = [f for f in os.listdir(tmp_path) if f.startswith("brk_expr_") and f.endswith(".co")]
    assert first[0] == 0 and len(files) == 1 and os.path.getsize(tmp_path / files[0]) > 4000
    second = run()
    assert second == (1, first[1])
    (tmp_path / files[0]).write_bytes(b"not a code object")
    third = run()
    assert third == (0, first[1]) and os.path.getsize(tmp_path / files[0]) > 4000


def test_apply_script_dispatch(oracle, dev):
    """apply_script: a script that IS one of the reference's stencils runs its tuned kernel, the same script with another constant is
    compiled (into the tuned star kernel), a 19-point off-axis script is compiled onto the cube tile, 343 taps go to the tap table,
    a non-linear script is compiled -- each against the oracle"""
    from test_spec import star_script
    from bricklib_amd.spec import load_spec
    ex = os.path.join(ROOT, "examples")
    grid, info = bl.init_grid((6, 6, 6), dev)
    ids = np.ascontiguousarray(grid.reshape(6, 6, 6)[1:-1, 1:-1, 1:-1]).reshape(-1)
    ids_dev = grid_to_device(ids, dev)
    st, so = (bl.BrickStorage.allocate(info.nbricks, 512, dev) for _ in range(2))
    st.dat.uniform_(-0.5, 0.5)
    hin = st.dat.cpu().numpy()
    A, B = bl.Brick(info, st, 0), bl.Brick(info, so, 0)
    coeff = oracle.coeffs(129, 1)

    def check(path, want_path, kernel_prefix, want):
        assert path == want_path and bl.last_kernel().startswith(kernel_prefix), (path, bl.last_kernel())
        assert np.abs(so.dat.cpu().numpy() - want).max() <= TOL * 4 and np.abs(want).max() > 0.01
        so.dat.zero_()

    def by_taps(spec, consts):
        want = np.zeros_like(hin)
        oracle.brick_taps(info.adj_host, hin, want, 512, ids, spec.taps(consts))
        return want
    from bricklib_amd.spec import MPI_DEFAULTS
    s13 = load_spec(star_script(2, ["MPI_B0", "MPI_B1", "MPI_B2"]))
    want = np.zeros_like(hin)
    oracle.brick_stencil(ST_MPI13, info.adj_host, hin, want, 512, ids=ids)
    check(bl.apply_script(s13, A, B, ids=ids_dev), "builtin:mpi13pt", "k_star_", want)
    other = dict(MPI_DEFAULTS, MPI_B1=0.08)
    check(bl.apply_script(s13, A, B, ids=ids_dev, constants={"MPI_B1": 0.08}), "compiled", "k_expr_jit<2,star_wg>", by_taps(s13, other))
    s7 = load_spec(star_script(1, ["coeff[0]", "coeff[1]"]).replace("c[1] * src(i + 1, j, k)", "c[1] * src(i + 1, j, k)"))
    check(bl.apply_script(s7, A, B, ids=ids_dev, constants={"coeff": coeff}), "compiled", "k_expr_jit<1,s7,list>", by_taps(s7, {"coeff": coeff}))
    seven = load_spec(EXPR_HEAD + """
u = Grid("bIn", 3); v = Grid("bOut", 3)
c = [ConstRef("coeff[%d]" % q) for q in range(7)]
v(i, j, k).assign(c[0] * u(i, j, k) + c[1] * u(i + 1, j, k) + c[2] * u(i - 1, j, k) + c[3] * u(i, j + 1, k) + c[4] * u(i, j - 1, k) +
                  c[5] * u(i, j, k + 1) + c[6] * u(i, j, k - 1))
STENCIL = [v]
""")
    want = np.zeros_like(hin)
    oracle.brick_stencil(ST_7PT_COEFF, info.adj_host, hin, want, 512, ids=ids, coeff=coeff)
    check(bl.apply_script(seven, A, B, ids=ids_dev, constants={"coeff": coeff}), "builtin:7pt", "k_s7_wg<kCoeff", want)
    lap = load_spec(os.path.join(ex, "lap19.py"))
    check(bl.apply_script(lap, A, B, ids=ids_dev, constants={"coeff": coeff}), "compiled", "k_expr_jit<1,cube_dense>", by_taps(lap, {"coeff": coeff}))
    blur = load_spec(os.path.join(ex, "blur343.py"))
    check(bl.apply_script(blur, A, B, ids=ids_dev, constants={"coeff": coeff}), "taps", "k_taps_far<3,2>", by_taps(blur, {"coeff": coeff}))
    mm = load_spec(os.path.join(ex, "minmod.py"))
    want = np.zeros_like(hin)
    oracle.brick_expr(mm, info.adj_host, hin, want, 512, ids, {"coeff": coeff})
    check(bl.apply_script(mm, A, B, brick_range=None, ids=ids_dev, constants={"coeff": coeff}), "compiled", "k_expr_jit<1,s7,list>", want)
    # the box form: the tuned box kernels for a built-in, brk_expr_apply_box for a compiled script
    box = (0, (6, 6, 6), (1, 1, 1), (4, 4, 4))
    check(bl.apply_script(mm, A, B, box=box, constants={"coeff": coeff}), "compiled", "k_expr_jit<1,s7,list>", want)
    want = np.zeros_like(hin)
    oracle.brick_stencil(ST_7PT_COEFF, info.adj_host, hin, want, 512, ids=ids, coeff=coeff)
    check(bl.apply_script(seven, A, B, box=box, constants={"coeff": coeff}), "builtin:7pt", "k_s7_wg<kCoeff,8,kModeBox>", want)
    with pytest.raises(ValueError):
        bl.apply_script(blur, A, B, box=box, constants={"coeff": coeff})


@pytest.mark.parametrize("case", ["cond", "star2", "star4", "limiter", "mixed1"])
def test_compiled_forms_over_irregular_lists(dev, case):
    """the tuned kernels a program is compiled into share halos between the waves of a workgroup WHEN consecutive list entries are
    i-neighbours; a shuffled list with skipped entries, a list running against i, and a twisted adjacency table (bricks whose -i / +i
    links do not match) take their fallback paths -- same bits as the interpreter, which works brick by brick"""
    from bricklib_amd.order import LIST_SKIP
    from bricklib_amd.spec import load_spec
    scripts = dict(EXPR_SCRIPTS, mixed1=EXPR_HEAD + """
hi = Func("max", 2)
u = Grid("u", 3); v = Grid("v", 3)
uxy = u(i + 1, j + 1, k) - u(i + 1, j - 1, k) - u(i - 1, j + 1, k) + u(i - 1, j - 1, k)
v(i, j, k).assign(u(i, j, k) + 0.25 * hi(uxy, u(i, j - 1, k + 1)) * If(uxy > 0, 1.0, 0.5))
STENCIL = [v]
""")
    spec = load_spec(scripts[case])
    consts = {"coeff": [0.1 * q - 0.25 for q in range(7)], "cfl": 0.3}
    shape = (11, 6, 5)
    grid, info = bl.init_grid(shape, dev)
    g3 = grid.reshape(5, 6, 11)
    inner = np.ascontiguousarray(g3[1:-1, 1:-1, 1:-1]).reshape(-1)
    rng = np.random.default_rng(9)
    shuffled = rng.permutation(inner).astype(np.uint32)
    shuffled[::7] = LIST_SKIP
    backwards = inner[::-1].copy()
    # a twisted table: two bricks swap their +i neighbours (the links between list neighbours no longer hold there)
    adj = info.adj_host.copy()
    a, b = int(g3[2, 2, 3]), int(g3[2, 3, 3])
    adj[a, 14], adj[b, 14] = adj[b, 14], adj[a, 14]
    twisted = bl.BrickInfo(adj, dev)
    st = bl.BrickStorage.allocate(info.nbricks, 512, dev)
    st.dat.uniform_(-0.5, 0.5)
    ck = bl.CompiledExpr(spec, consts)
    for binfo, lst in ((info, shuffled), (info, backwards), (twisted, inner), (twisted, shuffled)):
        s1, s2 = (bl.BrickStorage.allocate(info.nbricks, 512, dev) for _ in range(2))
        ids = grid_to_device(lst, dev)
        bl.apply_expr(spec, bl.Brick(binfo, st, 0), bl.Brick(binfo, s1, 0), ids=ids, constants=consts, interpret=True)
        ck.apply(bl.Brick(binfo, st, 0), bl.Brick(binfo, s2, 0), ids=ids)
        assert torch.equal(s1.dat, s2.dat) and s1.dat.abs().max().item() > 0.1
    ck.release()


@pytest.mark.parametrize("case", ["cond", "star2", "limiter", "far3"])
def test_compiled_expression_over_a_box(dev, case):
    """brk_expr_apply_box: the reference's launch shape (interior box of an init_grid numbering) for a compiled script -- the box is
    walked in the built-in stencils' order, no id list to build; same bits as the list form over the same bricks, nothing outside the
    box is written, geometry errors are refused"""
    from bricklib_amd._lib import BrkError
    from bricklib_amd.spec import load_spec
    spec = load_spec(EXPR_SCRIPTS[case])
    consts = {"coeff": [0.1 * q - 0.25 for q in range(7)], "cfl": 0.3}
    ck = bl.CompiledExpr(spec, consts)
    for shape in ((10, 10, 10), (7, 21, 5), (66, 13, 4)):
        ni, nj, nk = shape
        grid, info = bl.init_grid(shape, dev)
        st, s1, s2 = (bl.BrickStorage.allocate(info.nbricks, 512, dev) for _ in range(3))
        st.dat.uniform_(-0.5, 0.5)
        box = (0, shape, (1, 1, 1), (ni - 2, nj - 2, nk - 2))
        ck.apply(bl.Brick(info, st, 0), bl.Brick(info, s1, 0), box=box)
        ids = np.ascontiguousarray(grid.reshape(nk, nj, ni)[1:-1, 1:-1, 1:-1]).reshape(-1)
        ck.apply(bl.Brick(info, st, 0), bl.Brick(info, s2, 0), ids=grid_to_device(ids, dev))
        assert torch.equal(s1.dat, s2.dat) and s1.dat.abs().max().item() > 0.1
        mask = torch.ones(info.nbricks, dtype=torch.bool, device=dev)
        mask[torch.from_numpy(ids.astype(np.int64)).to(dev)] = False
        assert (s1.dat.view(-1, 512)[mask] == 0).all()
        ck.apply(bl.Brick(info, st, 0), bl.Brick(info, s1, 0), box=(0, shape, (1, 1, 1), (ni - 2, 0, nk - 2)))     # empty box
        with pytest.raises(BrkError):
            ck.apply(bl.Brick(info, st, 0), bl.Brick(info, s1, 0), box=(0, shape, (1, 1, 1), (ni, nj - 2, nk - 2)))
    s3 = bl.BrickStorage.allocate(info.nbricks, 512, dev)
    bl.apply_expr(spec, bl.Brick(info, st, 0), bl.Brick(info, s3, 0), box=box, constants=consts)      # the one-call form
    assert torch.equal(s3.dat, s1.dat)
    with pytest.raises(ValueError):
        bl.apply_expr(spec, bl.Brick(info, st, 0), bl.Brick(info, s3, 0), box=box, constants=consts, interpret=True)
    ck.release()


def test_expression_argument_checks(dev):
    """a program is checked before anything is launched: unknown codes, registers out of range or read before written, taps
    outside the 125-point cube, too many operations"""
    from bricklib_amd._lib import BrkError, ExprOp, lib
    from bricklib_amd.brick import _ptr
    from bricklib_amd.spec import EX
    grid, info = bl.init_grid((4, 4, 4), dev)
    st, so = (bl.BrickStorage.allocate(info.nbricks, 512, dev) for _ in range(2))

    def run(rows, result=0, n=1):
        tab = (ExprOp * max(1, len(rows)))()
        for q, (code, dst, a, b, c, off, imm) in enumerate(rows):
            tab[q].code, tab[q].dst, tab[q].a, tab[q].b, tab[q].c, tab[q].imm = code, dst, a, b, c, imm
            tab[q].offset[0], tab[q].offset[1], tab[q].offset[2] = off
        return lib.brk_stencil_apply_expr(_ptr(info.adj), _ptr(st.dat), _ptr(so.dat), 512, None, 0, n, tab, len(rows), result, None)
    tap = (EX["tap"], 0, 0, 0, 0, (0, 0, 0), 0.0)
    assert run([tap]) == 0
    assert run([tap], n=0) == 0
    for bad, result in (([], 0), ([(99, 0, 0, 0, 0, (0, 0, 0), 0.0)], 0), ([(EX["tap"], 16, 0, 0, 0, (0, 0, 0), 0.0)], 0),
                        ([(EX["add"], 0, 1, 2, 0, (0, 0, 0), 0.0)], 0), ([tap, (EX["add"], 1, 0, 3, 0, (0, 0, 0), 0.0)], 1),
                        ([(EX["tap"], 0, 0, 0, 0, (5, 0, 0), 0.0)], 0), ([tap], 5), ([tap] * 121, 0), ([tap], 16)):
        assert run(bad, result) != 0
    from bricklib_amd.spec import load_spec
    with pytest.raises(BrkError):       # in == out
        bl.apply_expr(load_spec(EXPR_SCRIPTS["cond"]), bl.Brick(info, st, 0), bl.Brick(info, st, 0), brick_range=(0, 1),
                      constants={"coeff": [0.1] * 7}, interpret=True)


def test_taps_argument_checks(dev):
    from bricklib_amd._lib import BrkError
    from bricklib_amd.stencils import apply_taps
    grid, info = bl.init_grid((4, 4, 4), dev)
    st, so = (bl.BrickStorage.allocate(info.nbricks, 512, dev) for _ in range(2))
    for bad in ([((5, 0, 0), 1.0)], [((3, 1, -5), 1.0)], [((0, 0, 0), 1.0)] * 126, [((4, 0, 0), 1.0)] * 730):
        with pytest.raises(BrkError):
            apply_taps(bad, bl.Brick(info, st, 0), bl.Brick(info, so, 0), brick_range=(0, 1))


def test_entry_point_edge_cases(oracle, dev):
    """empty and degenerate inputs, and the error convention of the hot-path entry points (return code + brk_last_error, nothing
    written): empty list / empty range / empty box are no-ops; in == out, step < 512, reversed ranges, boxes outside the
    numbering, unknown stencils and missing coefficient vectors are refused."""
    from bricklib_amd._lib import lib
    from bricklib_amd.brick import _ptr
    grid, info = bl.init_grid((5, 5, 5), dev)
    st, so = (bl.BrickStorage.allocate(info.nbricks, 512, dev) for _ in range(2))
    st.dat.uniform_(0.0, 1.0)
    A, B = bl.Brick(info, st, 0), bl.Brick(info, so, 0)
    empty = torch.zeros(0, dtype=torch.int32, device=dev)
    for name in ("7pt", "mpi7pt", "mpi13pt", "mpi25pt", "27pt", "mpi125pt", "mpi7pt_x2"):
        coeff = oracle.coeffs(129, 1) if name in ("7pt", "27pt") else None
        bl.apply_stencil(name, A, B, brick_range=(7, 7), coeff=coeff)                       # empty range
        bl.apply_stencil(name, A, B, box=(0, (5, 5, 5), (1, 1, 1), (3, 0, 3)), coeff=coeff)   # empty box
        assert lib.brk_stencil_apply_list(bl.STENCILS[name], _ptr(info.adj), _ptr(st.dat), _ptr(so.dat), 512, None, 0, None, None) in (0, -1)
        # a single brick, through every form
        b = int(grid.reshape(5, 5, 5)[2, 2, 2])
        want = np.zeros(info.nbricks * 512)
        oracle.brick_stencil(NAMES[name[:-3] if name.endswith("_x2") else name], info.adj_host, st.dat.cpu().numpy(), want, 512,
                             ids=np.array([b], dtype=np.uint32), coeff=coeff)
        if not name.endswith("_x2"):
            for kw in (dict(brick_range=(b, b + 1)), dict(ids=grid_to_device(np.array([b], dtype=np.uint32), dev)),
                       dict(box=(0, (5, 5, 5), (2, 2, 2), (1, 1, 1)))):
                so.dat.zero_()
                bl.apply_stencil(name, A, B, coeff=coeff, **kw)
                got = so.dat.cpu().numpy()
                assert np.abs(got - want).max() <= TOL, (name, kw.keys())
    assert (so.dat.view(-1, 512)[0] == 0).all()
    so.dat.zero_()
    U3 = C.c_uint32 * 3
    cases = [
        lambda: lib.brk_stencil_apply_range(bl.BRK_MPI7, _ptr(info.adj), _ptr(st.dat), _ptr(st.dat), 512, 1, 2, None, None),     # in == out
        lambda: lib.brk_stencil_apply_range(bl.BRK_MPI7, _ptr(info.adj), _ptr(st.dat), _ptr(so.dat), 256, 1, 2, None, None),     # step < 512
        lambda: lib.brk_stencil_apply_range(bl.BRK_MPI7, _ptr(info.adj), _ptr(st.dat), _ptr(so.dat), 512, 5, 2, None, None),     # reversed
        lambda: lib.brk_stencil_apply_range(bl.BRK_MPI7, _ptr(info.adj), C.c_void_p(st.dat.data_ptr() + 8), _ptr(so.dat), 512, 1, 2, None, None),  # field not 16-byte aligned
        lambda: lib.brk_stencil_apply_range(bl.BRK_MPI7, _ptr(info.adj), _ptr(st.dat), _ptr(so.dat), 513, 1, 2, None, None),      # odd step
        lambda: lib.brk_stencil_apply_range(bl.BRK_7PT_COEFF, _ptr(info.adj), _ptr(st.dat), _ptr(so.dat), 512, 1, 2, None, None),  # no coefficients
        lambda: lib.brk_stencil_apply_range(42, _ptr(info.adj), _ptr(st.dat), _ptr(so.dat), 512, 1, 2, None, None),               # unknown stencil
        lambda: lib.brk_stencil_apply_list(bl.BRK_MPI7, _ptr(info.adj), _ptr(st.dat), _ptr(so.dat), 512, None, 3, None, None),     # null list
        lambda: lib.brk_stencil_apply_box(bl.BRK_MPI7, _ptr(info.adj), _ptr(st.dat), _ptr(so.dat), 512, 0, U3(5, 5, 5), U3(3, 3, 3), U3(3, 1, 1), None, None),
        lambda: lib.brk_stencil_apply_range(bl.BRK_MPI7, None, _ptr(st.dat), _ptr(so.dat), 512, 1, 2, None, None),                # null adjacency
        lambda: lib.brk_link_copy(None, 3, 4096, None),
        lambda: lib.brk_exchange_view_create(None, _ptr(st.dat), C.byref(C.c_void_p())),
    ]
    for q, f in enumerate(cases):
        assert f() < 0 and len(lib.brk_last_error()) > 0, q
    torch.cuda.synchronize()
    assert (so.dat == 0).all()                      # nothing was launched by a refused call


def test_box_list_cache_replacement(oracle, dev):
    """more box geometries than the library caches id lists for (8): entries are replaced and rebuilt correctly"""
    shape = (12, 11, 10)
    grid, info = bl.init_grid(shape, dev)
    st = bl.BrickStorage.allocate(info.nbricks, 512, dev)
    st.dat.uniform_(0.0, 1.0)
    hin = st.dat.cpu().numpy()
    g3 = grid.reshape(shape[2], shape[1], shape[0])
    boxes = [((1, 1, 1), (3 + q % 5, 2 + q % 4, 1 + q % 3)) for q in range(11)]
    for rnd in range(2):
        for lo, cnt in boxes:
            so = bl.BrickStorage.allocate(info.nbricks, 512, dev)
            bl.apply_stencil("mpi13pt", bl.Brick(info, st, 0), bl.Brick(info, so, 0), box=(0, shape, lo, cnt))
            ids = np.ascontiguousarray(g3[lo[2]:lo[2] + cnt[2], lo[1]:lo[1] + cnt[1], lo[0]:lo[0] + cnt[0]]).reshape(-1)
            want = np.zeros_like(hin)
            oracle.brick_stencil(NAMES["mpi13pt"], info.adj_host, hin, want, 512, ids=ids)
            assert np.abs(so.dat.cpu().numpy() - want).max() <= TOL, (rnd, lo, cnt)


def test_shift_exchange_single_rank_equals_direct(oracle, dev):
    """ShiftExchanger on the GPU (one periodic rank: every stage is device links) fills the ghosts exactly like the
    run-by-run exchange; multi-rank routing is covered on CPU tensors in tests/test_exchange_gloo.py."""
    from bricklib_amd.decomp import Exchanger, ShiftExchanger
    d = bl.BrickDecomp((40, 24, 32))
    bl.populate(d, [1, 1, 1], [0, 0, 0])
    base = torch.arange(d.nbricks * 512, dtype=torch.float64)
    res = []
    for cls in (Exchanger, ShiftExchanger):
        st = bl.BrickStorage(base.clone().to(dev), d.nbricks, 512)
        cls(d, st, 0).exchange()
        torch.cuda.synchronize()
        res.append(st.dat.cpu())
    assert torch.equal(res[0], res[1])
    od = oracle.decomp((40, 24, 32))
    want = [base.numpy().copy()]
    oracle.exchange(od, want, [oracle.populate_rank_map([1, 1, 1], [0, 0, 0])], 512)
    assert (res[1].numpy() == want[0]).all()


@pytest.mark.parametrize("name", ["mpi7pt", "7pt"])
def test_fused_double_sweep(oracle, dev, name):
    """BRK_MPI7_X2 / BRK_7PT_X2: one pass = two 7-point sweeps.  Against two separate GPU sweeps (same bricks, the first over
    a ring more) and against the oracle's two array sweeps; list with skipped entries and BrickDecomp range incl. ghost."""
    from bricklib_amd.order import LIST_SKIP, tiled
    coeff = oracle.coeffs(129, 1) if name == "7pt" else None
    N = 40                                               # cells; array with 16 cells of padding+ghost per side as in single_setup
    grid, grid_dev, bInfo, arr, st, bIn, ids = single_setup(oracle, dev, N, 512)
    SB = (N + 16) // 8
    g3 = grid.reshape(SB, SB, SB)
    inner = tiled(g3[2:-2, 2:-2, 2:-2])                  # bricks whose 2-cell neighbourhood lies inside the filled region
    ring = tiled(g3[1:-1, 1:-1, 1:-1])
    mid, two, fused = (bl.BrickStorage.allocate(bInfo.nbricks, 512, dev) for _ in range(3))
    B = lambda s: bl.Brick(bInfo, s, 0)
    bl.apply_stencil(name, bIn, B(mid), ids=grid_to_device(ring, dev), coeff=coeff)
    bl.apply_stencil(name, B(mid), B(two), ids=grid_to_device(inner, dev), coeff=coeff)
    bl.apply_stencil(name + "_x2", bIn, B(fused), ids=grid_to_device(inner, dev), coeff=coeff)
    real = torch.from_numpy(inner[inner != LIST_SKIP].astype(np.int64)).to(dev)
    a, b = two.dat.view(-1, 512)[real], fused.dat.view(-1, 512)[real]
    assert (a - b).abs().max().item() <= TOL
    mask = torch.ones(bInfo.nbricks, dtype=torch.bool, device=dev)
    mask[real] = False
    assert (fused.dat.view(-1, 512)[mask] == 0).all()    # bricks outside the list are untouched
    # oracle: two array sweeps
    a1, a2 = np.zeros_like(arr), np.zeros_like(arr)
    oracle.array_stencil(NAMES[name], arr, a1, (9,) * 3, (arr.shape[0] - 9,) * 3, coeff=coeff)
    oracle.array_stencil(NAMES[name], a1, a2, (24,) * 3, (N + 8,) * 3, coeff=coeff)   # bricks 2..SB-3 = array cells 24..N+8
    got = torch.zeros(arr.shape, dtype=torch.float64, device=dev)
    bl.copyFromBrick((N + 16,) * 3, (8,) * 3, (0,) * 3, got, grid_dev, B(fused))
    sl = (slice(24, N + 8),) * 3
    assert np.abs(got.cpu().numpy()[sl] - a2[sl]).max() <= TOL
    # range form over a BrickDecomp numbering incl. the ghost ring: equals two sweeps wherever two sweeps are defined
    d = bl.BrickDecomp((32, 24, 40))
    dinfo = d.getBrickInfo(dev)
    A, M, T2, F = (dinfo.allocate(512, dev) for _ in range(4))
    A.dat.uniform_(0.0, 1.0)
    A.dat[:512] = 0
    Bd = lambda s: bl.Brick(dinfo, s, 0)
    bl.apply_stencil(name, Bd(A), Bd(M), brick_range=(1, d.sep_pos[2]), coeff=coeff)
    bl.apply_stencil(name, Bd(M), Bd(T2), brick_range=(1, d.sep_pos[1]), coeff=coeff)
    bl.apply_stencil(name + "_x2", Bd(A), Bd(F), brick_range=(1, d.sep_pos[1]), coeff=coeff)
    assert (T2.dat[512:512 * d.sep_pos[1]] - F.dat[512:512 * d.sep_pos[1]]).abs().max().item() <= TOL


@pytest.mark.timeout(600)
def test_bench_line_contract_small():
    """bench.py end to end at a small size: ONE JSON line with the contract's keys (metric, value, steps, roofline with
    bound/achieved/peak/frac, cpu_baseline, config with a workload name), the round-2 additions (timing windows, steady state,
    config3 block with per-stencil roofline and oracle error), and a sampled oracle error within the parity bar."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cells", "128", "--steps", "5", "--warmup", "2", "--windows", "3",
                        "--steady-seconds", "0.2", "--cpu-seconds", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=560)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["dtype"] == "f64" and d["config"]["workload"] == "single-7pt-128"
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    # at 128^3 the reference leg (built for 512^3) is not available: the brick-layout port is the baseline of the line
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert len(d["timing"]["window_ms_per_step"]) == 3 and d["steady_state"]["launches"] > 0
    assert d["max_abs_err_vs_oracle"] <= TOL
    for nm in ("27pt", "mpi125pt", "mpi13pt", "mpi25pt"):
        assert d["config3"][nm]["max_abs_err_vs_oracle"] <= TOL and d["config3"][nm]["roofline"]["frac"] > 0
        assert d["config3"][nm]["kernel_ms_one_way"] == d["config3"][nm]["kernel_ms"] and d["config3"][nm]["kernel_ms_alternating"] > 0
    assert d["weak_workload_on_this_gpu"]["value"] > 0
    # round 5: stencil scripts given as data -- the tuned kernel, the run-time compiled kernel (star-shaped, radius 1: k_s7_wg around the
    # program) and the interpreter, and the fp64-vector roofline beside the HBM one
    sc = d["stencil_scripts"]
    assert sc["compiled_at_run_time"]["kernel"] == "k_expr_jit<1,s7,list>" and sc["interpreted"]["kernel"] == "k_expr<1,4>"
    assert sc["compiled_at_run_time"]["kernel_ms"] < sc["interpreted"]["kernel_ms"] and sc["tuned_d3cond_kernel"]["GStencil/s"] > 0
    assert d["config3"]["mpi125pt"]["roofline_fp64_vector"]["flop_per_point"] == 134
    # round 4: every sweep walks the bricks in the same direction by default (value / roofline.frac are HBM numbers); the launches
    # with alternating directions (Infinity-Cache re-reads of the input) are timed beside the headline, --alternate swaps the roles
    assert d["config"]["sweep_direction"] == "one way" and d["alternating"]["kernel_ms"] > 0 and d["alternating"]["value"] > 0 and "one_way" not in d
    assert d["roofline"]["traffic"] is None       # the stored PMC measurement is for 512^3
    alt, _ = _bench_line(["--cells", "64", "--steps", "3", "--warmup", "1", "--windows", "1", "--steady-seconds", "0", "--no-cpu-baseline", "--no-config3",
                          "--no-weak-reference", "--alternate"])
    assert alt["config"]["sweep_direction"] == "alternating" and alt["one_way"]["kernel_ms"] > 0 and "alternating" not in alt


def _bench_line(args, timeout=560):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    return json.loads(lines[0]), r.stderr.decode()


@pytest.mark.timeout(600)
def test_bench_gpus2_starts_its_own_ranks_weak():
    """`python bench.py --gpus 2` from plain python (no launcher around it): bench.py starts its two ranks as a child
    torch.distributed.run (the `mpirun -np N weak/cuda` of weak/main.cu:46-60) -- here with --backend gloo, two processes sharing
    the one GPU, packed ghost bytes staged through host memory -- and rank 0 prints ONE line that describes itself: the GPUs
    seen, the like-for-like one-GPU base, the exchange self-check, the conservation property of the periodic global field."""
    d, err = _bench_line(["--gpus", "2", "--backend", "gloo", "--cells", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert "starting 2 ranks" in err
    c = d["config"]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and c["workload"].startswith("weak-mpi7pt-64") and c["cart"] == [2, 1, 1]
    assert len(c["ranks_seen"]) == 2 and {x["rank"] for x in c["ranks_seen"]} == {0, 1}
    assert d["scaling_base"]["value"] > 0 and d["scaling_efficiency"] > 0
    assert c["conservation_rel_drift"] <= 1e-12
    assert c["exchange_self_check"]["passed"] and c["transport"] == "gloo-staged" and c["messages_per_exchange"] >= 1
    # round 4: the reference's per-phase lines (weak/main.cpp:271-289) in the JSON line, {min, avg, max, sigma} over the ranks
    ph = c["phases"]
    for k in ("pack_ms", "transport_ms", "unpack_ms", "calc_ms", "host_call_ms"):
        assert set(ph[k]) == {"min", "avg", "max", "sigma"} and ph[k]["min"] <= ph[k]["avg"] <= ph[k]["max"], (k, ph[k])
    assert ph["calc_ms"]["avg"] > 0 and ph["transport_ms"]["avg"] > 0 and ph["cycle_ms_timed"] > 0 and ph["exposed_exchange_ms"] >= 0
    assert "calc [" in err and "transport [" in err and "σ" in err


@pytest.mark.timeout(600)
def test_bench_gpus2_starts_its_own_ranks_strong():
    """the same for the strong workload (strong/main.cu:397-510 shape): global 128^3 in 32^3 subdomains over two ranks"""
    d, _ = _bench_line(["--gpus", "2", "--backend", "gloo", "--workload", "strong", "--strong-dom", "128", "--strong-sub", "32", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and c["workload"] == "strong-mpi13pt-128-32" and c["subdomains_per_rank"] == 32
    assert c["conservation_rel_drift"] <= 1e-12 and c["remote_bytes_per_rank"] > 0
    assert c["phases"]["calc_ms"]["avg"] > 0 and c["phases"]["pack_ms"]["max"] >= c["phases"]["pack_ms"]["min"] > 0


@pytest.mark.timeout(600)
def test_bench_transport_auto_falls_back_to_torch():
    """--transport auto when the C-ABI communicator cannot be created (test hook: brk_comm_unique_id made to fail, what a missing
    librccl looks like): the run moves to the torch transport -- batch_isend_irecv over RCCL on tensors that alias the view's
    buffers -- says why, passes the exchange self-check and conserves the field.  One GPU: the self-neighbour runs of the 1x1x1
    periodic rank go through the communicator (--self-via-comm) instead of device links, so the transport really moves the bytes."""
    base = ["--workload", "weak", "--cells", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--self-via-comm", "--schedule", "plain"]
    d, err = _bench_line(base + ["--transport", "auto", "--test-fail-comm"])
    c = d["config"]
    assert c["transport"] == "torch" and "test-fail-comm" in c["transport_fallback_reason"] and "torch transport" in err
    assert c["exchange_self_check"]["passed"] and c["exchange_self_check"]["messages"] >= 1 and c["conservation_rel_drift"] <= 1e-12
    d2, _ = _bench_line(base + ["--transport", "auto"])
    c2 = d2["config"]
    assert c2["transport"].startswith("abi") and c2["transport_fallback_reason"] is None and c2["exchange_self_check"]["passed"]
    # ONE RCCL in the process: the C-ABI transport binds to the library torch has already mapped (round 4, brk_comm_library)
    import os
    assert c2["rccl_library"] and "rccl" in c2["rccl_library"]
    assert os.path.realpath(c2["rccl_library"]).startswith(os.path.realpath(os.path.dirname(torch.__file__))) or "rocm" in c2["rccl_library"], c2["rccl_library"]
    assert c2["conservation_rel_drift"] <= 1e-12 and c2["messages_per_exchange"] == c["messages_per_exchange"]


@pytest.mark.timeout(900)
def test_weak_cycle_baseline_size_512(oracle, dev):
    """BASELINE config 4 at its real size on one GPU: ONE periodic 1x1x1 rank of 512^3 (BrickDecomp, region-ordered ids, 287 497
    bricks, 42 ghost runs = 104 MB per exchange), one brick_func cycle (weak/main.cpp:172-213) of the 7-point stencil as 8 sweeps
    and as 4 fused double sweeps, under every schedule (plain / surface overlap / deep overlap) and with the exchange done by
    device links and through the 1-rank RCCL communicator (pack -> ncclSend/ncclRecv -> unpack, C ABI).
    Checks: (1) all schedule x transport combinations of one stencil give the same field BIT FOR BIT; (2) the periodic global
    sum is conserved to 1e-12; (3) the whole subdomain equals the oracle's cycle -- oracle exchange, 7 sweeps over [1,sep2) and
    one over [1,sep1) on the same bricks -- within the parity bar; (4) fused and plain cycles agree to rounding."""
    from bricklib_amd.decomp import Comm, Exchanger
    from bricklib_amd.drivers import SweepLists, TrapezoidLists, weak_cycle_deep
    N = 512
    d = bl.BrickDecomp((N, N, N))
    rmap = bl.populate(d, [1, 1, 1], [0, 0, 0])
    assert d.sep_pos == [238329, 262145, 287497] and len(d.ghost) == 42
    info = d.getBrickInfo(dev)
    s0, s1, s2 = d.sep_pos
    torch.manual_seed(5)
    init = torch.rand(d.nbricks * 512, dtype=torch.float64, device=dev)
    init[512 * s1:] = -7.0                    # ghosts: junk until exchanged
    init[:512] = 0.0                          # brick 0: the padding brick every out-of-domain neighbour points to
    stA, stB = info.allocate(512, dev), info.allocate(512, dev)
    A, B = bl.Brick(info, stA, 0), bl.Brick(info, stB, 0)
    lists = SweepLists(d, dev)
    main, xs = torch.cuda.current_stream(), torch.cuda.Stream()
    comm = Comm.init_rank(Comm.unique_id(), 0, 1)
    sum0 = float(init[512:512 * s1].sum().item())
    results = {}
    try:
        for name in ("mpi7pt", "mpi7pt_x2"):
            S = bl.stencil_iters(name)
            deep = TrapezoidLists(d, S, dev)
            ref = None
            for via in ("links", "rccl"):
                ex = Exchanger(d, stA, 0, mode="packed", comm=comm if via == "rccl" else None, self_via_comm=via == "rccl")
                assert (ex.messages >= 1) == (via == "rccl")
                for schedule in ("plain", "surface", "deep"):
                    stA.dat.copy_(init)
                    stB.dat.zero_()
                    if schedule == "deep":
                        weak_cycle_deep(name, d, A, B, ex.exchange, deep, main=main, comm=xs, state={})
                    else:
                        weak_cycle(name, d, A, B, ex.exchange, main=main, comm=xs, overlap=schedule == "surface", lists=lists, state={})
                    torch.cuda.synchronize()
                    got = stA.dat[512:512 * s1]
                    drift = abs(float(got.sum().item()) - sum0) / abs(sum0)
                    assert drift <= 1e-12, (name, via, schedule, drift)
                    if ref is None:
                        ref = got.clone()
                    else:
                        assert torch.equal(ref, got), (name, via, schedule)
                del ex
            results[name] = ref
    finally:
        torch.cuda.synchronize()
        comm.destroy()
    assert (results["mpi7pt"] - results["mpi7pt_x2"]).abs().max().item() <= TOL
    # the oracle's cycle on the same bricks
    cur = init.cpu().numpy()
    maps = [oracle.populate_rank_map([1, 1, 1], [0, 0, 0])]
    od = oracle.decomp((N, N, N))
    oracle.exchange(od, [cur], maps, 512)
    nxt = np.zeros_like(cur)
    for sweep in range(8):
        last = sweep == 7
        oracle.brick_stencil(ST_MPI7, d.adj, cur, nxt, 512, n=(s1 if last else s2) - 1, first=1)
        cur, nxt = nxt, cur
    want = torch.from_numpy(cur[512:512 * s1]).to(dev)
    for name in results:
        err = (results[name] - want).abs().max().item()
        assert err <= TOL, (name, err)


@pytest.mark.timeout(900)
def test_strong_cycle_baseline_shape_512_128(oracle, dev):
    """BASELINE config 5's shape on one GPU at half the edge: global 512^3 in 64 subdomains of 128^3 (Z-Morton ids, 18^3 bricks
    each, one launch per sweep over all of them), 13-point stencil, one exchange + ST_ITER = 4 sweeps (strong/main.cu:397-510).
    Checks: conservation of the periodic global sum to 1e-12, and one whole subdomain plus the sampled corners of all the others
    against the global periodic array swept by the oracle (the reference's own -v validation compares against exactly that,
    strong/main.cpp:359-365,402-464)."""
    from bricklib_amd.strong import StrongRank
    dom, sub = 512, 128
    sr = StrongRank(dom, sub, 0, 1, dev)
    nper = dom // sub
    assert sr.nsub == nper ** 3 and sr.peers == []
    # the field as a global periodic array, cut into the subdomains' interiors (ghosts arrive by the exchange)
    rng = np.random.default_rng(9)
    g = rng.random((dom, dom, dom))
    view = sr.storages[0].dat.view(sr.nsub, sr.nb, 512)
    view.zero_()
    for q in range(sr.nsub):
        x, y, z = sr.origin(q)            # global cell coordinate (i, j, k) of subdomain slot q (Z-Morton id q)
        blk = np.zeros((sub + 16,) * 3)
        blk[8:-8, 8:-8, 8:-8] = g[z:z + sub, y:y + sub, x:x + sub]
        bl.copyToBrick((sub + 16,) * 3, (0,) * 3, (0,) * 3, torch.from_numpy(blk).to(dev), sr.grid_dev, sr.sub_brick(0, q))
    sum0 = float(view[:, 1:sr.d.sep_pos[1]].sum().item())
    assert abs(sum0 - g.sum()) <= 1e-9 * g.sum()
    sr.exchange(None, torch.cuda.current_stream())      # one rank: every ghost run is aliased to its source brick, nothing moves
    sr.sweeps("mpi13pt")
    torch.cuda.synchronize()
    view = sr.storages[0].dat.view(sr.nsub, sr.nb, 512)
    drift = abs(float(view[:, 1:sr.d.sep_pos[1]].sum().item()) - sum0) / sum0
    assert drift <= 1e-12, drift
    # oracle: 4 sweeps of the global periodic array (wrap-padded by 8 = 4 sweeps x radius 2)
    cur = np.pad(g, 8, mode="wrap")
    for s in range(1, 5):
        nxt = np.zeros_like(cur)
        oracle.array_stencil(ST_MPI13, cur, nxt, (2 * s,) * 3, tuple(dom + 16 - 2 * s for _ in range(3)))
        cur = nxt
    core = cur[8:-8, 8:-8, 8:-8]
    for q in range(sr.nsub):
        x, y, z = sr.origin(q)
        got = torch.zeros((sub + 16,) * 3, dtype=torch.float64, device=dev)
        bl.copyFromBrick((sub,) * 3, (0,) * 3, (8,) * 3, got, sr.grid_dev, sr.sub_brick(0, q))
        want = core[z:z + sub, y:y + sub, x:x + sub]
        gg = got[8:-8, 8:-8, 8:-8]
        if q % 9 == 0:                 # every ninth subdomain whole, of the others the 8^3 corner blocks
            err = np.abs(gg.cpu().numpy() - want).max()
        else:
            err = max(np.abs(gg[a:a + 8, b:b + 8, c:c + 8].cpu().numpy() - want[a:a + 8, b:b + 8, c:c + 8]).max()
                      for a in (0, sub - 8) for b in (0, sub - 8) for c in (0, sub - 8))
        assert err <= TOL, (q, err)


@pytest.mark.timeout(900)
def test_strong_cycle_full_baseline_1024_128(oracle, dev):
    """BASELINE config 5 at its FULL size on one GPU (the share of all 8 ranks): global 1024^3 in 512 subdomains of 128^3, 13-point
    stencil, one exchange + ST_ITER = 4 sweeps (strong/main.cu:397-510).  The input is a hash of the GLOBAL periodic cell
    coordinate, so any neighbourhood can be rebuilt on the host: six sampled subdomains -- the corners of the Z-Morton cube that
    wrap around the periodic domain, and inner ones -- are compared cell by cell with 4 oracle sweeps of their (128 + 16)^3
    neighbourhood (what the reference's -v validation does for every subdomain, strong/main.cpp:359-365,402-464), and the sum of
    the whole field must be conserved to 1e-12."""
    from bricklib_amd.strong import StrongRank
    dom, sub = 1024, 128
    free = torch.cuda.mem_get_info()[0]
    if free < 48 * 2 ** 30:
        pytest.skip("needs 48 GiB of free device memory")
    sr = StrongRank(dom, sub, 0, 1, dev)
    assert sr.nsub == 512 and sr.peers == []

    def field(xp, z, y, x):           # the same integers on the device (torch) and on the host (numpy)
        h = (x * 73856093) ^ (y * 19349663) ^ (z * 83492791)
        h = ((h ^ (h >> 13)) & 0x7FFFFFFF) * 1274126177      # < 2^62: no wrap-around in either library
        return ((h ^ (h >> 16)) & 0xFFFFF).astype(np.float64) / 1048576.0 if xp is np else ((h ^ (h >> 16)) & 0xFFFFF).to(torch.float64) / 1048576.0

    view = sr.storages[0].dat.view(sr.nsub, sr.nb, 512)
    view.zero_()
    ar = torch.arange(sub, device=dev, dtype=torch.int64)
    blk = torch.zeros((sub + 16,) * 3, dtype=torch.float64, device=dev)
    for q in range(sr.nsub):
        x, y, z = sr.origin(q)
        blk[8:-8, 8:-8, 8:-8] = field(torch, (z + ar)[:, None, None], (y + ar)[None, :, None], (x + ar)[None, None, :])
        bl.copyToBrick((sub + 16,) * 3, (0,) * 3, (0,) * 3, blk, sr.grid_dev, sr.sub_brick(0, q))
    sum0 = float(view[:, 1:sr.d.sep_pos[1]].sum().item())
    sr.exchange(None, torch.cuda.current_stream())
    sr.sweeps("mpi13pt")
    torch.cuda.synchronize()
    view = sr.storages[0].dat.view(sr.nsub, sr.nb, 512)
    assert abs(float(view[:, 1:sr.d.sep_pos[1]].sum().item()) - sum0) / sum0 <= 1e-12
    coords = np.arange(-8, sub + 8, dtype=np.int64)
    for q in (0, 7, 73, 255, 438, 511):
        x, y, z = sr.origin(q)
        cur = field(np, ((z + coords) % dom)[:, None, None], ((y + coords) % dom)[None, :, None], ((x + coords) % dom)[None, None, :])
        cur = np.ascontiguousarray(np.broadcast_to(cur, (sub + 16,) * 3))
        for s_ in range(1, 5):
            nxt = np.zeros_like(cur)
            oracle.array_stencil(ST_MPI13, cur, nxt, (2 * s_,) * 3, tuple(sub + 16 - 2 * s_ for _ in range(3)))
            cur = nxt
        got = torch.zeros((sub + 16,) * 3, dtype=torch.float64, device=dev)
        bl.copyFromBrick((sub,) * 3, (0,) * 3, (8,) * 3, got, sr.grid_dev, sr.sub_brick(0, q))
        err = np.abs(got[8:-8, 8:-8, 8:-8].cpu().numpy() - cur[8:-8, 8:-8, 8:-8]).max()
        assert err <= TOL, (q, err)
    del sr, view
    torch.cuda.empty_cache()


def test_registered_adjacency_box_form_is_bit_identical(oracle, dev):
    """brk_adjacency_register (the BrickInfo of init_grid / getBrickInfo registers its device table): the 7-point box kernels then
    skip the adjacency rows of bricks whose face neighbours are the predicted ones.  Same bits as the unregistered path -- also
    where prediction fails: the inner box of a BrickDecomp (its border bricks point into the region-ordered surface), a periodic
    wrap written into the table, and after release / re-registration of a rewritten table."""
    from bricklib_amd._lib import check, lib
    import ctypes as C
    rng = np.random.default_rng(4)
    for case in ("init_grid", "decomp_inner", "wrapped"):
        if case == "decomp_inner":
            d = bl.BrickDecomp((48, 40, 56))
            adj = d.adj.copy()
            nb = [x // 8 - 2 for x in (48, 40, 56)]
            box = (1, nb, (0, 0, 0), nb)
        else:
            shape = (12, 9, 7)
            grid, info0 = bl.init_grid(shape, None)
            adj = info0.adj_host.copy()
            if case == "wrapped":       # i-periodic interior: the +i neighbour of the last interior column is the first one
                g3 = grid.reshape(shape[2], shape[1], shape[0])
                for k in range(shape[2]):
                    for j in range(shape[1]):
                        adj[g3[k, j, shape[0] - 2], 14] = g3[k, j, 1]
                        adj[g3[k, j, 1], 12] = g3[k, j, shape[0] - 2]
            box = (0, shape, (1, 1, 1), tuple(x - 2 for x in shape))
        nbricks = adj.shape[0]
        stA = bl.BrickStorage.allocate(nbricks, 512, dev)
        stA.dat.copy_(torch.from_numpy(rng.random(nbricks * 512)))
        outs = []
        raw = grid_to_device(adj.reshape(-1), dev).reshape(-1, 27)           # a device tensor: not registered
        for info in (bl.BrickInfo(raw), bl.BrickInfo(adj, dev)):               # unregistered, registered
            for name in ("7pt", "mpi7pt"):
                stB = bl.BrickStorage.allocate(nbricks, 512, dev)
                bl.apply_stencil(name, bl.Brick(info, stA, 0), bl.Brick(info, stB, 0), box=box, coeff=oracle.coeffs(129, 1) if name == "7pt" else None)
                outs.append(stB.dat.clone())
        assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3]), case
        assert outs[0].abs().max().item() > 0
    # list form of the cube kernels: a registered table replaces the adjacency rows of the list neighbours by two bits per brick
    # (linked to the -i / +i neighbour); tiled, lexicographic, ragged and shuffled lists, also over the wrapped table above
    ids_all = np.ascontiguousarray(grid.reshape(shape[2], shape[1], shape[0])[1:-1, 1:-1, 1:-1]).reshape(-1)
    from bricklib_amd.order import tiled
    lists = [tiled(grid.reshape(shape[2], shape[1], shape[0])[1:-1, 1:-1, 1:-1]), ids_all, ids_all[3:-5:2].copy(), rng.permutation(ids_all).astype(np.uint32)]
    for name in ("27pt", "mpi125pt", "mpi7pt_x2", "7pt", "mpi7pt", "mpi13pt", "mpi25pt"):   # star kernels: compact face table
        for ids in lists:
            outs = []
            for info in (bl.BrickInfo(raw), bl.BrickInfo(adj, dev)):
                stB = bl.BrickStorage.allocate(nbricks, 512, dev)
                bl.apply_stencil(name, bl.Brick(info, stA, 0), bl.Brick(info, stB, 0), ids=grid_to_device(ids, dev), coeff=oracle.coeffs(129, 1) if name in ("27pt", "7pt") else None)
                outs.append(stB.dat.clone())
            assert torch.equal(outs[0], outs[1]), (name, ids.size)
            assert outs[0].abs().max().item() > 0
    # the promise is per table: duplicate registration and unknown release are errors; release, rewrite, register again
    t = grid_to_device(adj.reshape(-1), dev)
    check(lib.brk_adjacency_register(C.c_void_p(t.data_ptr()), nbricks))
    assert lib.brk_adjacency_register(C.c_void_p(t.data_ptr()), nbricks) != 0
    check(lib.brk_adjacency_release(C.c_void_p(t.data_ptr())))
    assert lib.brk_adjacency_release(C.c_void_p(t.data_ptr())) != 0


@pytest.mark.parametrize("shape", [(10, 5, 6), (18, 12, 7), (34, 4, 11), (66, 19, 5)])
def test_125pt_four_waves_per_simd_kernel(oracle, dev, shape):
    """k_cube125_q (brk_cube125q.hip, round 5): the 125-point stencil through the box entry point over a REGISTERED table whose box is
    fully regular and has rows of a multiple of 8 bricks runs the four-waves-per-SIMD kernel (exact 40 960-byte group layout, corner
    cells in registers, shift slots aliased into the tiles) -- the oracle's result (stencils/fake.h:87-341), both sweep directions, with
    neighbour bricks that differ in every halo region (random field incl. the ghost ring).  The same box over an unregistered table,
    over a table with one irregular brick, and a box whose rows are not a multiple of 8 bricks keep k_cube_dense; all agree to rounding."""
    ni, nj, nk = shape
    grid, info0 = bl.init_grid(shape, None)
    adj = info0.adj_host.copy()
    nbricks = adj.shape[0]
    rng = np.random.default_rng(21)
    st = bl.BrickStorage.allocate(nbricks, 512, dev)
    st.dat.copy_(torch.from_numpy(rng.random(nbricks * 512)))
    hin = st.dat.cpu().numpy()
    g3 = grid.reshape(nk, nj, ni)
    ids = np.ascontiguousarray(g3[1:-1, 1:-1, 1:-1]).reshape(-1)
    want = np.zeros_like(hin)
    oracle.brick_stencil(ST_MPI125, adj, hin, want, 512, ids=ids)
    box = (0, shape, (1, 1, 1), (ni - 2, nj - 2, nk - 2))
    reg = bl.BrickInfo(adj, dev)                                             # registers its device table
    for reverse in (False, True, False):
        out = bl.BrickStorage.allocate(nbricks, 512, dev)
        bl.apply_stencil("mpi125pt", bl.Brick(reg, st, 0), bl.Brick(reg, out, 0), box=box, reverse=reverse)
        assert bl.last_kernel().startswith("k_cube125_q"), bl.last_kernel()
        got = out.dat.cpu().numpy()
        assert np.abs(got - want).max() <= TOL, (shape, reverse, np.abs(got - want).max())
    # fallbacks: unregistered table; one irregular brick (a corner entry of an interior brick points at the dummy brick); ragged rows
    raw = bl.BrickInfo(grid_to_device(adj.reshape(-1), dev).reshape(-1, 27))
    out = bl.BrickStorage.allocate(nbricks, 512, dev)
    bl.apply_stencil("mpi125pt", bl.Brick(raw, st, 0), bl.Brick(raw, out, 0), box=box)
    assert bl.last_kernel().startswith("k_cube_dense"), bl.last_kernel()
    assert np.abs(out.dat.cpu().numpy() - want).max() <= TOL
    adj2 = adj.copy()
    adj2[g3[nk // 2, nj // 2, ni // 2], 26] = 0
    want2 = np.zeros_like(hin)
    oracle.brick_stencil(ST_MPI125, adj2, hin, want2, 512, ids=ids)
    twisted = bl.BrickInfo(adj2, dev)
    out = bl.BrickStorage.allocate(nbricks, 512, dev)
    bl.apply_stencil("mpi125pt", bl.Brick(twisted, st, 0), bl.Brick(twisted, out, 0), box=box)
    assert bl.last_kernel().startswith("k_cube_dense"), bl.last_kernel()
    assert np.abs(out.dat.cpu().numpy() - want2).max() <= TOL
    if ni - 2 > 8:
        box7 = (0, shape, (2, 1, 1), (ni - 3, nj - 2, nk - 2))                # rows of ni - 3 bricks: not a multiple of 8
        ids7 = np.ascontiguousarray(g3[1:-1, 1:-1, 2:-1]).reshape(-1)
        want7 = np.zeros_like(hin)
        oracle.brick_stencil(ST_MPI125, adj, hin, want7, 512, ids=ids7)
        out = bl.BrickStorage.allocate(nbricks, 512, dev)
        bl.apply_stencil("mpi125pt", bl.Brick(reg, st, 0), bl.Brick(reg, out, 0), box=box7)
        assert bl.last_kernel().startswith("k_cube_dense"), bl.last_kernel()
        assert np.abs(out.dat.cpu().numpy() - want7).max() <= TOL


@pytest.mark.parametrize("shape", [(10, 10, 10), (7, 21, 5), (14, 6, 9), (66, 11, 4)])
def test_star_two_bricks_per_wave_kpaired_lists(oracle, dev, shape):
    """k_star_kp (two bricks per wave; the planes between the two bricks of a k-pair are loaded once): the 25-point star over
    k-paired lists (brk_stencil_apply_list_kpaired; order.tiled(kpair=True)) with odd level counts, rows that are not multiples
    of 4 and fewer rows than XCDs, and -- every list DECLARED k-paired, which selects the pair kernel for both stars whatever the
    list really is -- over lists that are NOT paired: tiled, lexicographic, ragged, shuffled.  Always the oracle's result."""
    from bricklib_amd._lib import lib
    from bricklib_amd.order import tiled
    ni, nj, nk = shape
    grid, info = bl.init_grid(shape, dev)
    g3 = grid.reshape(nk, nj, ni)[1:-1, 1:-1, 1:-1]
    rng = np.random.default_rng(12)
    st = bl.BrickStorage.allocate(info.nbricks, 512, dev)
    st.dat.copy_(torch.from_numpy(rng.random(info.nbricks * 512)))
    hin = st.dat.cpu().numpy()
    flat = np.ascontiguousarray(g3).reshape(-1)
    lists = {"kpaired": tiled(g3, kpair=True), "kpaired4": tiled(g3, rows_per_block=4, kpair=True), "tiled": tiled(g3), "lexi": flat,
             "ragged": flat[1:-3:3].copy(), "shuffled": rng.permutation(flat).astype(np.uint32)}
    for name, sid in (("mpi25pt", ST_MPI25), ("mpi13pt", ST_MPI13)):
        want = {}
        for lname, ids in lists.items():
            real = ids[ids != 0xFFFFFFFF]
            w = np.zeros_like(hin)
            oracle.brick_stencil(sid, info.adj_host, hin, w, 512, ids=np.ascontiguousarray(real))
            want[lname] = w
        for declared in (False, True):
            for lname, ids in lists.items():
                out = bl.BrickStorage.allocate(info.nbricks, 512, dev)
                bl.apply_stencil(name, bl.Brick(info, st, 0), bl.Brick(info, out, 0), ids=grid_to_device(ids, dev), kpaired=declared or lname.startswith("kpaired"))
                err = np.abs(out.dat.cpu().numpy() - want[lname]).max()
                assert err <= TOL, (name, declared, lname, err)
