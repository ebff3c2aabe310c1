"""Placement of brick storages in HBM (brk_storage_alloc / brk_placement_probe, include/brk.h; bricklib_amd.BrickStorage.allocate).
Needs an MI355X.  What is held here: a placed storage is an ordinary tensor over memory the library owns, results do not depend on
where a storage lies, the memory goes back when the tensor dies, and the measurement behind the choice is a real one."""
import ctypes as C
import gc

import numpy as np
import pytest
import torch

import bricklib_amd as bl
from bricklib_amd._lib import Placement, lib
from bricklib_amd.brick import grid_to_device

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def small_storages_are_measured(monkeypatch):
    """the library measures storages of 512 MiB and more (a probe over less runs out of the Infinity Cache); these tests exercise the
    MACHINERY on storages of 87-128 MiB, so the threshold is lowered for them (BRK_PLACEMENT_MIN_BYTES, read at every call)"""
    monkeypatch.setenv("BRK_PLACEMENT_MIN_BYTES", str(64 << 20))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_placed_storages_are_tensors_and_results_do_not_depend_on_placement(dev, monkeypatch):
    """a 22^3-brick grid with step 1024 is 87 MB per storage: large enough to be measured (threshold lowered to 64 MiB by the fixture),
    small enough for a quick test"""
    SB = 22
    grid, info = bl.init_grid((SB, SB, SB), dev)
    ids = grid_to_device(np.ascontiguousarray(grid.reshape(SB, SB, SB)[1:-1, 1:-1, 1:-1]).reshape(-1), dev)
    outs = {}
    for placed in ("1", "0"):
        monkeypatch.setenv("BRK_PLACEMENT", placed)
        a, b = bl.BrickStorage.allocate(info.nbricks, 1024, dev), bl.BrickStorage.allocate(info.nbricks, 1024, dev)
        assert a.dat.dtype == torch.float64 and a.dat.numel() == info.nbricks * 1024 and a.dat.is_contiguous()
        assert float(a.dat.abs().max()) == 0.0 and float(b.dat.abs().max()) == 0.0      # allocate() hands out zeroed storages
        if placed == "1":
            assert a.placement is not None and b.placement is not None
            assert b.placement["candidates"] >= 2 and 0 <= b.placement["chosen"] < b.placement["candidates"]
            assert 0.0 < b.placement["probe_sweep_ms_chosen"] <= b.placement["probe_sweep_ms_worst"]
            edge = int(round((info.nbricks * 2) ** (1 / 3))) + 1   # the probe sweeps the interior of the largest cubic grid of 4-KiB bricks that fits
            while edge ** 3 > info.nbricks * 2:
                edge -= 1
            assert b.placement["probe_bricks"] == (edge - 2) ** 3
        else:
            assert a.placement is None and b.placement is None
        g = torch.Generator(device=dev).manual_seed(7)
        a.dat.copy_(torch.rand(a.dat.numel(), generator=g, device=dev, dtype=torch.float64))
        for name in ("7pt", "mpi13pt", "mpi125pt"):
            coeff = np.linspace(0.1, 1.0, 129) if name == "7pt" else None
            bl.apply_stencil(name, bl.Brick(info, a, 0), bl.Brick(info, b, 0), ids=ids, coeff=coeff)
            torch.cuda.synchronize()
            outs[(placed, name)] = b.dat.clone()
        # views, slices and arithmetic behave as on any tensor
        assert bl.Brick(info, b, 512).dat.data_ptr() == b.dat.data_ptr() + 512 * 8
        del a, b
    for name in ("7pt", "mpi13pt", "mpi125pt"):
        assert torch.equal(outs[("1", name)], outs[("0", name)])


def test_placed_memory_is_returned_with_the_tensor(dev, monkeypatch):
    monkeypatch.setenv("BRK_PLACEMENT", "1")
    gc.collect()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    nbytes = 96 << 20
    sts = [bl.BrickStorage.allocate(nbytes // 8 // 512, 512, dev) for _ in range(3)]
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free0 - free1 >= 3 * nbytes * 0.95            # three storages are held, the rejected candidates are not
    assert free0 - free1 <= 3 * nbytes + (64 << 20)
    view = sts[1].dat[1000:2000]                         # a view keeps its storage alive
    del sts
    gc.collect()
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info(dev)
    assert free0 - free2 >= nbytes * 0.95 and free0 - free2 <= nbytes + (64 << 20)
    assert float(view.sum()) == 0.0
    del view
    gc.collect()
    torch.cuda.synchronize()
    free3, _ = torch.cuda.mem_get_info(dev)
    assert free0 - free3 <= (32 << 20)


def test_spacer_path_of_the_allocator(dev, monkeypatch):
    """when every back-to-back candidate runs at one rate the allocator moves on behind untouched spacer allocations; the test forces
    that path (BRK_PLACEMENT_TEST_SPACERS) and holds that it works and that everything but the chosen storage is given back"""
    monkeypatch.setenv("BRK_PLACEMENT", "1")
    monkeypatch.setenv("BRK_PLACEMENT_TEST_SPACERS", "1")
    gc.collect()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    nbytes = 96 << 20
    a = bl.BrickStorage.allocate(nbytes // 8 // 512, 512, dev)
    b = bl.BrickStorage.allocate(nbytes // 8 // 512, 512, dev)
    assert b.placement["spacer_gib"] >= 4 and b.placement["candidates"] >= 8
    assert 0 <= b.placement["chosen"] < b.placement["candidates"]
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free0 - free1 <= 2 * nbytes + (64 << 20)       # spacers and rejected candidates are gone
    a.dat.fill_(1.0)
    b.dat.copy_(a.dat)
    assert float(b.dat.sum()) == a.dat.numel()
    del a, b
    gc.collect()


def test_placement_probe_and_c_abi_allocator(dev):
    nbytes = 128 << 20
    a, b = C.c_void_p(), C.c_void_p()
    rep = Placement()
    with torch.cuda.device(dev):
        assert lib.brk_storage_alloc(nbytes, C.byref(a), C.byref(rep)) == 0 and a.value
        assert lib.brk_storage_alloc(nbytes, C.byref(b), C.byref(rep)) == 0 and b.value and b.value != a.value
        assert rep.candidates >= 2 and rep.ms_chosen > 0.0 and rep.probe_bricks == 30 ** 3      # 128 MiB = 32768 bricks = 32^3
        ms = C.c_double()
        assert lib.brk_placement_probe(a, b, nbytes, None, C.byref(ms)) == 0
        # 27000 bricks swept: 110 MB read + 110 MB written; between 2 and 12 TB/s that is 0.02-0.11 ms; anything from 0.01 to 1 ms is a measurement
        assert 0.01 < ms.value < 1.0
        assert abs(ms.value - rep.ms_chosen) / rep.ms_chosen < 0.3      # the same sweep between the same storages
        assert lib.brk_placement_probe(a, C.c_void_p(b.value + 8), nbytes - 4096, None, C.byref(ms)) != 0
        assert lib.brk_placement_probe(a, b, 26 * 4096, None, C.byref(ms)) != 0
        assert lib.brk_storage_free(a) == 0 and lib.brk_storage_free(a) != 0
        assert lib.brk_storage_free(b) == 0


def test_default_threshold_leaves_small_storages_alone(dev, monkeypatch):
    """without the override a storage below 512 MiB is a plain hipMalloc: candidates = 0, no transient allocations"""
    monkeypatch.delenv("BRK_PLACEMENT_MIN_BYTES")
    rep = Placement()
    a, b = C.c_void_p(), C.c_void_p()
    with torch.cuda.device(dev):
        assert lib.brk_storage_alloc(128 << 20, C.byref(a), C.byref(rep)) == 0 and rep.candidates == 0
        assert lib.brk_storage_alloc(128 << 20, C.byref(b), C.byref(rep)) == 0 and rep.candidates == 0
        assert lib.brk_storage_free(a) == 0 and lib.brk_storage_free(b) == 0
    st = bl.BrickStorage.allocate((128 << 20) // 8 // 512, 512, dev)
    assert st.placement is None


def test_three_field_chain_with_named_partners(dev, monkeypatch):
    """brk_storage_alloc_ex at the size the library measures by itself (three storages of 1.07 GiB, the 512^3 field): a sweep
    out = f(a, b) wants its two READ fields of the same kind and the output of the other.  a is the first storage (nothing to measure);
    b is placed with {a, SAME}; out with {a, OPPOSITE}, {b, OPPOSITE}.  Held here: every call measures (candidates >= 2), the chosen
    output is not slower than the slowest candidate seen, the sweep a -> out is not slower than the same-allocation rate (the slow
    rate of the device) by more than the noise, wrong partners are refused, and the default chain (A B A) still alternates."""
    monkeypatch.delenv("BRK_PLACEMENT_MIN_BYTES")
    from bricklib_amd._lib import PlacePartner
    nb = 66 ** 3
    nbytes = nb * 4096
    free0, _ = torch.cuda.mem_get_info(dev)
    if free0 < 40 * nbytes:
        pytest.skip("not enough free device memory for the candidates")
    a = bl.BrickStorage.allocate(nb, 512, dev)
    assert a.placement is not None and a.placement["candidates"] == 0          # first storage of its size: no partner
    b = bl.BrickStorage.allocate(nb, 512, dev, partners=[(a, "same")])
    out = bl.BrickStorage.allocate(nb, 512, dev, partners=[(a, "opposite"), (b, "opposite")])
    for st in (b, out):
        assert st.placement["candidates"] >= 2 and st.placement["probe_sweep_ms_same_allocation"] > 0.0
    assert out.placement["probe_sweep_ms_chosen"] <= out.placement["probe_sweep_ms_worst"]
    ms_ab, ms_ao, ms_bo = C.c_double(), C.c_double(), C.c_double()
    with torch.cuda.device(dev):
        scratch = torch.empty(nb * 512, dtype=torch.float64, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        assert lib.brk_placement_probe(p(a.dat), p(out.dat), nbytes, None, C.byref(ms_ao)) == 0
        assert lib.brk_placement_probe(p(b.dat), p(out.dat), nbytes, None, C.byref(ms_bo)) == 0
        del scratch
    slow = out.placement["probe_sweep_ms_same_allocation"]
    assert ms_ao.value <= slow * 1.01 and ms_bo.value <= slow * 1.01, (ms_ao.value, ms_bo.value, slow)
    # refused: a partner that is not a placed storage, an unknown relation
    t = torch.zeros(1 << 20, dtype=torch.float64, device=dev)
    tab = (PlacePartner * 1)()
    tab[0].storage_dev, tab[0].relation = t.data_ptr(), 0
    ptr = C.c_void_p()
    with torch.cuda.device(dev):
        assert lib.brk_storage_alloc_ex(nbytes, tab, 1, None, C.byref(ptr), None) != 0 and b"did not come from" in lib.brk_last_error()
        tab[0].storage_dev, tab[0].relation = a.dat.data_ptr(), 7
        assert lib.brk_storage_alloc_ex(nbytes, tab, 1, None, C.byref(ptr), None) != 0
    del b, out
    gc.collect()
    # the default chain: each storage against its predecessor as a read / write pair
    c1 = bl.BrickStorage.allocate(nb, 512, dev)
    c2 = bl.BrickStorage.allocate(nb, 512, dev)
    assert c1.placement["candidates"] >= 2 and c2.placement["candidates"] >= 2
    del a, c1, c2
    gc.collect()
    torch.cuda.synchronize()
