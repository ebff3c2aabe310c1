"""N>1 path on CPU: two (and four) processes, backend gloo, exercising the same Exchanger plan the GPU path uses
(grouped isend/irecv per run + local links for self-neighbours) and checking it against the oracle's exchange."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, dom, q, mode, order="good"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bricklib_amd as bl
        from bricklib_amd.decomp import Exchanger, rank_to_coo
        cart = bl.dims_create(world)
        d = bl.BrickDecomp(dom, order=order)
        bl.populate(d, cart, rank_to_coo(rank, cart))
        st = bl.BrickStorage(torch.zeros(d.nbricks * 512, dtype=torch.float64), d.nbricks, 512)
        # every element encodes (rank, brick, element) so a misrouted run is caught exactly
        v = torch.arange(d.nbricks * 512, dtype=torch.float64) + rank * 1e9
        st.dat.copy_(v)
        if mode == "shift":
            from bricklib_amd.decomp import ShiftExchanger
            ex = ShiftExchanger(d, st, rank, allow_cpu_plumbing=True)
            ex.exchange()
            dist.barrier()
            assert ex.messages <= 6
            q.put((rank, st.dat.numpy().copy(), ex.messages))
            return
        ex = Exchanger(d, st, rank, allow_cpu_plumbing=True, mode=mode)
        ex.exchange()
        dist.barrier()
        q.put((rank, st.dat.numpy().copy(), len(ex.remote)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,order", [(2, "direct", "good"), (4, "direct", "good"), (2, "packed", "good"), (4, "packed", "good"), (8, "packed", "good"),
                                              (8, "direct", "good"), (2, "shift", "good"), (4, "shift", "good"), (8, "shift", "good"), (3, "shift", "good"),
                                              (3, "packed", "good"), (6, "direct", "good"),
                                              (4, "packed", "bad"), (2, "direct", "normal"), (4, "shift", "bad")])   # the reference's other section orders
def test_exchange_two_ranks_gloo(oracle, world, mode, order):
    dom = (32, 16, 24)
    port = 29500 + (os.getpid() % 2000) + world + {"direct": 0, "packed": 10, "shift": 20}[mode] + {"good": 0, "normal": 40, "bad": 50}[order]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dom, q, mode, order)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, arr, nremote = q.get(timeout=120)
        got[r] = (arr, nremote)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # oracle: same initial storages, exchange in one address space
    import bricklib_amd as bl
    from bricklib_amd.decomp import rank_to_coo
    od = oracle.decomp(dom, order=order)
    cart = bl.dims_create(world)
    storages = [np.arange(od.nbricks * 512, dtype=np.float64) + r * 1e9 for r in range(world)]
    maps = [oracle.populate_rank_map(cart, rank_to_coo(r, cart)) for r in range(world)]
    oracle.exchange(od, storages, maps, 512)
    for r in range(world):
        assert (got[r][0] == storages[r]).all(), r
        assert got[r][1] > 0        # some runs really went over the wire


def _bootstrap_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        from bricklib_amd.decomp import Comm
        calls = {"init": 0}

        def no_id():
            raise RuntimeError("librccl could not be loaded (test)")

        def no_init(uid, r, n):
            calls["init"] += 1
            raise RuntimeError("ncclCommInitRank must not be reached")

        Comm.unique_id = staticmethod(no_id)          # what rank 0 sees when RCCL is missing; the other ranks never call it
        Comm.init_rank = classmethod(lambda cls, uid, r, n: no_init(uid, r, n))
        comm, why = Comm.bootstrap_all_or_none()
        # the ranks are still in step: a collective right behind the decision completes on all of them
        t = torch.tensor([rank + 1])
        dist.all_reduce(t)
        q.put((rank, comm is None, why, int(t.item()), calls["init"]))
    finally:
        dist.destroy_process_group()


def test_comm_bootstrap_failure_is_collective():
    """rank 0 failing to create the RCCL id must not leave the other ranks inside the broadcast (mismatched collectives,
    hang until the watchdog).  Every rank gets (None, reason) from bootstrap_all_or_none and the group is still usable."""
    world = 3
    port = 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bootstrap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(g[0] for g in got) == [0, 1, 2]
    for rank, none, why, total, inits in got:
        assert none and "librccl could not be loaded (test)" in why and total == 6 and inits == 0


def test_bench_gpus_n_without_gpus_fails_fast():
    """`python bench.py --gpus 2` on a host without two GPUs: one clear line and a non-zero status before any rank is started
    (the parent never initialises the GPU runtime)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                       env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    if torch.cuda.device_count() >= 2:
        pytest.skip("this host has two GPUs")
    assert r.returncode == 2 and b"needs 2 visible GPUs" in r.stderr and r.stdout.strip() == b""


def test_bench_help_lists_every_option():
    """`python bench.py --help` formats (argparse expands `%` in help strings) and names the options the docs quote"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0, r.stderr.decode()[-400:]
    for opt in ("--gpus", "--steps", "--warmup", "--workload", "--schedule", "--exchange", "--transport", "--backend", "--strong-local-ghosts"):
        assert opt in out, opt
