#!/usr/bin/env python3
"""bench.py -- headline benchmark: GStencil/s (lattice updates/s) of the 7-pt fp64 brick sweep, 512^3 per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: starts its N ranks itself, as a child torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1  workload "single-7pt-512": the reference's single/hip d3pt7 `Trans` launch (stencils/3axis.hip.cpp:165-171):
       lexicographic 66^3 brick grid, 7pt.py coefficients, one step = one sweep over the 64^3 interior bricks; input
       and output fields in two storages (--layout interleaved = one storage as in single/, 1.5 % slower).
N > 1  workload "weak-mpi7pt-512": weak/main.cu shape: BrickDecomp 512^3 per rank on a periodic MPI_Dims_create(N) grid,
       one step = one brick_func = ghost exchange (per-peer pack -> RCCL send/recv -> unpack, device links for
       self-neighbours) + ST_ITER = 8 sweeps -- by default as 4 passes over memory of 2 sweeps each (--fuse) -- the exchange
       hidden behind the part of the passes that cannot depend on it yet (--schedule).  GStencil/s counts useful updates
       only (weak/main.cpp:285-287).
Other workloads: --workload weak at N = 1, --workload strong (global 1024^3 in 128^3 subdomains, 13-point).
Prints ONE JSON line on rank 0.  Inputs are synthetic U[0,1) already resident in HBM when timing starts.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL / device-memory sharing fail with the legacy mode)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0
BYTES_PER_UPDATE = 16.0        # one fp64 read + one fp64 write per lattice update
FP64_VECTOR_TFLOPS = 78.65     # /opt/skills/guides/MI355X_MICROARCH.md: fp32 vector 157.3 TFLOP/s; fp64 FMA issues at half that rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--n", "--cells", dest="n", type=int, default=512, help="cells per axis per GPU (default 512 = BASELINE config)")
    ap.add_argument("--stencil", default=None, help="override stencil (7pt|27pt|mpi7pt|mpi13pt|mpi25pt|mpi125pt)")
    ap.add_argument("--variant", type=int, default=0, help="experiments: kernel variant of the chosen stencil (brk_set_variant; 0 = default)")
    ap.add_argument("--workload", default=None, choices=[None, "single", "weak", "strong"],
                    help="single = 7-pt sweep of one 512^3 grid (default at 1 GPU); weak = 512^3 per rank with ghost exchange "
                         "(default at N>1); strong = fixed global --strong-dom^3 cut into --strong-sub^3 subdomains (config 5)")
    ap.add_argument("--strong-dom", type=int, default=1024)
    ap.add_argument("--strong-sub", type=int, default=128)
    ap.add_argument("--strong-local-ghosts", choices=["aliased", "copied"], default="aliased",
                    help="strong workload, ghost runs whose source subdomain lives on the same rank: aliased = reached through the "
                         "adjacency table, neither copied nor swept (default; same subdomain cells bit for bit); copied = device "
                         "links as in the reference (strong/main.cu:201-208)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-weak-reference", action="store_true", help="single workload: skip the extra (untimed) weak-workload line")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--schedule", default="auto", choices=("auto", "deep", "surface", "plain"),
                    help="weak workload: deep = exchange hidden behind the early passes of all sweeps, surface = behind "
                         "the inner bricks of one sweep, plain = in order, auto = cheapest of the three for the measured exchange time")
    ap.add_argument("--exchange", default="packed", choices=("packed", "direct", "shift"),
                    help="weak workload, runs that leave the GPU: packed = one RCCL send+recv per peer through pack/unpack "
                         "kernels, direct = one per run straight from the storage (the reference's zero-copy form), shift = "
                         "three-stage shift exchange, <= 2 messages per stage (MultiStageExchangeView)")
    ap.add_argument("--transport", default="auto", choices=("auto", "abi", "torch"),
                    help="N > 1, messages that leave the GPU: abi = ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd issued inside libbrk.so "
                         "(brk_exchange_transport) on a communicator created through the C ABI, bootstrapped over the torch.distributed "
                         "group; torch = the same messages as torch.distributed batch_isend_irecv (backend nccl = RCCL); auto = abi, "
                         "unless creating that communicator fails on any rank (decided collectively, before any data moves): then torch")
    ap.add_argument("--self-via-comm", action="store_true",
                    help="weak workload on ONE GPU: send the self-neighbour runs through a 1-rank RCCL communicator (ncclSend/ncclRecv "
                         "to self inside libbrk.so) instead of device links -- exercises the whole pack -> RCCL -> unpack path, on the "
                         "exchange stream of the chosen schedule, where only one GPU is available")
    ap.add_argument("--test-fail-comm", action="store_true", help=argparse.SUPPRESS)   # tests: creating the C-ABI communicator fails
    ap.add_argument("--watchdog-seconds", type=float, default=900.0,
                    help="N > 1: a rank that has not finished after this long prints where it was and exits with status 3")
    ap.add_argument("--no-scaling-base", action="store_true", help="N > 1: skip the 1-GPU run of the same workload on rank 0")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl = RCCL over xGMI (one rank per GPU); gloo = REHEARSAL of the multi-rank control flow with several "
                         "ranks sharing one GPU, ghost bytes staged through host memory (numbers are not a measurement)")
    ap.add_argument("--form", default="box", choices=("box", "list"),
                    help="single workload: box = brk_stencil_apply_box over the interior of the init_grid numbering (the "
                         "reference's launch shape; the 7-point kernels derive the ids arithmetically), list = "
                         "brk_stencil_apply_list over the same bricks in the same order")
    ap.add_argument("--layout", default="separate", choices=("separate", "interleaved"),
                    help="single workload: separate = input and output fields in two storages (step 512, the layout of the "
                         "reference's weak/strong drivers); interleaved = both fields in one storage (step 1024, the layout of "
                         "single/: 3axis.cpp:25-28).  Same kernels; interleaving reads and writes in the same DRAM pages costs "
                         "1.5 %% (7-pt) to 10 %% (27-pt) on MI355X (profiles/r01_h_layouts.json)")
    ap.add_argument("--fuse", type=int, default=2, choices=(1, 2),
                    help="weak workload with the 7-point stencil: 2 = every pass over memory performs two of the ST_ITER = 8 sweeps "
                         "between exchanges (BRK_MPI7_X2, the 25-point S*S stencil; results equal to rounding), 1 = one sweep per pass")
    ap.add_argument("--deep-depth", type=int, default=0,
                    help="--schedule deep: split only the first K sweeps into early/late passes (0 = all ST_ITER sweeps)")
    ap.add_argument("--emulate-exchange-ms", type=float, default=0.0,
                    help="EXPERIMENT (one-GPU boxes): add a spin kernel of this length to every exchange, on the exchange's "
                         "stream, to see how much exchange time each schedule hides; results are not a measurement of any link")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--windows", type=int, default=5, help="single workload: the K-step timed region is repeated this many times; the median is reported")
    ap.add_argument("--steady-seconds", type=float, default=2.0, help="single workload: extra steady-state figure over this many seconds of launches (0 = skip)")
    ap.add_argument("--alternate", action="store_true",
                    help="headline with successive sweeps over the same bricks in ALTERNATING directions (BRK_SWEEP_REVERSE on every other "
                         "launch: same bricks, same results): a sweep then starts in the part of the never-rewritten input the 256 MB "
                         "Infinity Cache still holds from the one before, so part of the reads is not served by HBM.  Default (round 4): "
                         "every sweep walks the bricks in the same direction -- value and roofline.frac are HBM numbers -- and the "
                         "alternating figure is reported beside them (`alternating` block)")
    ap.add_argument("--one-way", action="store_true", help="accepted for compatibility: one direction is the default")
    ap.add_argument("--weak-zigzag", action="store_true", help="weak workload: alternate the direction of successive sweeps too (no gain measured: "
                                                                "a ping-pong sweep reads what the previous one wrote with non-temporal stores)")
    ap.add_argument("--deep-early", choices=["auto", "0", "1"], default="auto",
                    help="weak workload, deep schedule: the last sweep writes the surface first and the next exchange starts under the rest of that pass "
                         "(+0.2 ms of exchange hidden for +2.5 %% of the cycle); auto = chosen with the schedule from the measured exchange time")
    ap.add_argument("--no-weak-prediction", action="store_true", help="single workload: skip the emulated-exchange rows (predicted_weak_efficiency) behind weak_workload_on_this_gpu")
    ap.add_argument("--no-scripts", action="store_true", help="single workload: skip the stencil-scripts-as-data lines (stencil_scripts: tuned / compiled / interpreted)")
    ap.add_argument("--no-config3", action="store_true", help="single workload: skip the 27-/125-/13-/25-point lines on the same grid")
    return ap.parse_args()


def verify_sample(bl, O, sid, info, st_in, st_out, step, off_out, ids, coeff, nsample=2048, seed=5):
    """size-independent spot check at full size: `nsample` random bricks of the sweep vs the CPU oracle."""
    rng = np.random.default_rng(seed)
    sub = np.sort(rng.choice(ids, size=min(nsample, ids.size), replace=False)).astype(np.uint32)
    hin = st_in.dat.cpu().numpy().reshape(-1, step)
    fin = np.ascontiguousarray(hin[:, :512]).reshape(-1) if step != 512 else hin.reshape(-1)
    want = np.zeros_like(fin)
    O.brick_stencil(sid, info.adj_host, fin, want, 512, ids=sub, coeff=coeff)
    got = st_out.dat.cpu().numpy().reshape(-1, step)[sub, off_out:off_out + 512]
    return float(np.abs(got - want.reshape(-1, 512)[sub]).max())


def usable_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box exposes every
    host thread in the mask but grants a 16-CPU share)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(O, sid, info, st, step, ids, coeff, n_cells, seconds):
    """The oracle (CPU restatement, OpenMP over bricks) on the same sweep, all usable host cores, timed with the reference's
    CPU protocol time_func (stencils/stencils.h:40-53): one warm-up call, then batches of 1, 2, 4, ... calls until TOT_TIME
    has passed; seconds per call = elapsed / (calls made).  TOT_TIME = `seconds` (the reference uses 5 s on a quiet node)."""
    from bricklib_amd.timing import time_func
    cores = usable_cores()
    O.L.orc_set_threads(cores)
    hin = st.dat.cpu().numpy().reshape(-1, step)
    fin = np.ascontiguousarray(hin[:, :512]).reshape(-1) if step != 512 else hin.reshape(-1)
    out = np.zeros_like(fin)
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    O.brick_stencil(sid, info.adj_host, fin, out, 512, ids=ids, coeff=coeff)       # first touch of `out` (page faults)
    per_call, calls, dt = time_func(lambda: O.brick_stencil(sid, info.adj_host, fin, out, 512, ids=ids, coeff=coeff), seconds)
    return {"value": n_cells / per_call / 1e9, "unit": "GStencil/s", "cores": cores, "kind": "port",
            "protocol": "time_func (stencils/stencils.h:40-53), TOT_TIME = %.0f s" % seconds,
            "sample": f"{calls} full sweeps of the same {round(n_cells ** (1 / 3))}^3 brick list with oracle/brick_oracle.c "
                      f"(OpenMP, {cores} threads), {dt:.1f} s"}


def cpu_baseline_reference(n_cells_axis, seconds, stencil="7pt", coeff=None):
    """The reference's OWN CPU code on this host: its `_TILEFOR` loop nest (stencils/stencils.h:30-38) around the arithmetic of
    the benchmarked stencil -- for the 7pt.py headline the seven-coefficient expression of d3pt7's arr_func
    (stencils/3axis.cpp:31-37), for mpi7pt its `ST_CPU` macro (stencils/fake.h:345-349) -- array layout, compiled in place into
    oracle/_ref/libbrickref.so (oracle/ref_cpu_sweep.cpp) with the reference's Release flags, under the reference's `time_func`.
    A stated baseline, never the target.  One call updates STRIDEG^3 = (N + 16)^3 points (the reference sweeps its ghost zone
    too); `value` counts only the N^3 useful points, `value_incl_ghost_zone` all of them."""
    import ctypes as C
    from bricklib_amd.timing import time_func
    path = os.path.join(ROOT, "oracle", "_ref", "libbrickref.so")
    if n_cells_axis != 512 or not os.path.exists(path):
        return None
    L = C.CDLL(path)
    if not hasattr(L, "ref_cpu_sweep_mpi7"):
        return None
    L.ref_cpu_sweep_stride.restype = L.ref_cpu_sweep_points.restype = C.c_long
    cores = usable_cores()
    L.ref_cpu_sweep_threads(cores)
    stride, points = int(L.ref_cpu_sweep_stride()), int(L.ref_cpu_sweep_points())
    a = np.random.default_rng(3).random(stride ** 3)
    b = np.zeros_like(a)
    pa, pb = a.ctypes.data, b.ctypes.data
    if stencil == "7pt" and hasattr(L, "ref_cpu_sweep_7pt"):
        c7 = np.ascontiguousarray((coeff if coeff is not None else np.random.default_rng(1).random(7))[:7], dtype=np.float64)
        L.ref_cpu_sweep_7pt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        pc = c7.ctypes.data
        call, what = (lambda: L.ref_cpu_sweep_7pt(pa, pb, pc)), "the seven-coefficient expression of d3pt7's arr_func (stencils/3axis.cpp:31-37)"
    else:
        L.ref_cpu_sweep_mpi7.argtypes = [C.c_void_p, C.c_void_p]
        call, what = (lambda: L.ref_cpu_sweep_mpi7(pa, pb)), "ST_CPU (stencils/fake.h:345-349: MPI_ALPHA / MPI_BETA)"
    per_call, calls, dt = time_func(call, seconds)
    return {"value": float(n_cells_axis) ** 3 / per_call / 1e9, "unit": "GStencil/s", "cores": cores,
            "kind": "reference", "code": "reference _TILEFOR (stencils/stencils.h:30-38) + " + what + ", array layout, compiled in place (oracle/ref_cpu_sweep.cpp)",
            "value_incl_ghost_zone": points / per_call / 1e9,
            "protocol": "time_func (stencils/stencils.h:40-53), TOT_TIME = %.0f s" % seconds,
            "sample": f"{calls} calls over a {stride}^3 array: {points} points updated per call incl. the ghost zone it also sweeps, "
                      f"{n_cells_axis}^3 of them useful (value counts only those), OpenMP {cores} threads, {dt:.1f} s"}


COND_SHAPED_SCRIPT = """
from st.expr import Index, ConstRef, If
from st.grid import Grid
from st.func import Func
i = Index(0); j = Index(1); k = Index(2)
pos = Func("max", 2)
u = Grid("bIn", 3); v = Grid("bOut", 3)
total = None
for q, (di, dj, dk) in ((5, (0, 0, 1)), (6, (0, 0, -1)), (3, (0, 1, 0)), (4, (0, -1, 0)), (1, (1, 0, 0)), (2, (-1, 0, 0)), (0, (0, 0, 0))):
    term = ConstRef("coeff[%d]" % q) * pos(u(i + di, j + dj, k + dk), 0.0)
    total = term if total is None else total + term
v(i, j, k).assign(If(total > 0, total, -total))
STENCIL = [v]
"""


def script_lines(bl, info, st, sto, step_e, off_out, ids, stream, launches=40):
    """ms per sweep of a cond-shaped script over the brick list `ids` (tiled order): tuned kernel, compiled at run time, interpreted"""
    from bricklib_amd.brick import grid_to_device
    from bricklib_amd.order import tiled
    from bricklib_amd.spec import load_spec
    spec = load_spec(COND_SHAPED_SCRIPT)
    coeff = np.random.default_rng(1).random(129) - 0.45
    g3 = np.asarray(ids).reshape(-1)
    n = round(g3.size ** (1.0 / 3.0))
    lst = grid_to_device(tiled(g3.reshape(n, n, n)) if n * n * n == g3.size else g3, st.dat.device)
    bIn, bOut = bl.Brick(info, st, 0), bl.Brick(info, sto, off_out)
    t0 = time.time()
    ck = bl.CompiledExpr(spec, {"coeff": coeff})
    compile_s = time.time() - t0
    out = {"script": "|sum_t c_t * max(in_t, 0)| over the 7-point star (the shape of stencils/cond.py), given as text at run time",
           "brk_expr_compile_s": compile_s, "from_disk_cache": ck.from_disk}

    def timed(fn, reps):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        cells = float(g3.size) * 512.0
        return {"kernel_ms": ms, "GStencil/s": cells / (ms * 1e-3) / 1e9, "roofline_frac": BYTES_PER_UPDATE * cells / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "kernel": bl.last_kernel()}
    out["tuned_d3cond_kernel"] = timed(lambda: bl.apply_stencil("cond", bIn, bOut, ids=lst, coeff=coeff, stream=stream), launches)
    out["compiled_at_run_time"] = timed(lambda: ck.apply(bIn, bOut, ids=lst, stream=stream), launches)
    out["interpreted"] = timed(lambda: bl.apply_expr(spec, bIn, bOut, ids=lst, constants={"coeff": coeff}, stream=stream, interpret=True), 5)
    ck.release()
    return out


def side_stencil(bl, O, name, info, st, sto, step_e, off_out, ids, box, stream, launches=100, nsample=512, warm_ms=150.0, zigzag=False):
    """One of the other stencils of BASELINE configs 2-3 on the SAME 512^3 grid through the same box entry point, outside
    the timed headline: kernel time from HIP events on the launch stream, roofline fraction, sampled oracle check."""
    from bricklib_amd.stencils import STENCILS
    sid = STENCILS[name]
    coeff = np.random.default_rng(1).random(129) if name in ("7pt", "27pt") else None
    bIn, bOut = bl.Brick(info, st, 0), bl.Brick(info, sto, off_out)
    zz = bl.Zigzag()

    def sweep(alternate=zigzag):
        bl.apply_stencil(sid, bIn, bOut, box=box, coeff=coeff, stream=stream, reverse=alternate and zz.turn())

    # the sampled oracle check of the previous stencil leaves the GPU idle for seconds: warm up until the clocks are back
    # (>= warm_ms of back-to-back launches) before the timed launches
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record(stream)
    for _ in range(20):
        sweep()
    w1.record(stream)
    torch.cuda.synchronize()
    for _ in range(int(warm_ms / max(w0.elapsed_time(w1) / 20, 1e-3))):
        sweep()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(stream)
    for _ in range(launches):
        sweep()
    e1.record(stream)
    for _ in range(launches):
        sweep(alternate=not zigzag)      # the other mode beside it
    e2.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / launches
    cells = float(len(ids)) * 512.0
    ach = BYTES_PER_UPDATE * cells / (ms * 1e-3) / 1e9
    other = e1.elapsed_time(e2) / launches
    r = {"kernel_ms": ms, "kernel_ms_one_way": other if zigzag else ms, "kernel_ms_alternating": ms if zigzag else other,
         "sweep_direction": "alternating" if zigzag else "one way", "GStencil/s": cells / (ms * 1e-3) / 1e9, "launches": launches,
         "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                      "frac_of_measured_copy": ach / HBM_COPY_GBS, "kernel": bl.last_kernel()}}
    # every stencil here is HBM-bound by arithmetic intensity; the 125-point one only with coefficient grouping (134 flop
    # per point, the reference's ST_CPU form) -- "report both rooflines for it".  Algorithmic flop per point, not instructions executed
    # (the kernels fold symmetric taps before multiplying); peak = the fp64 vector rate, half of the guide's 157.3 TFLOP/s fp32 vector.
    flop = {"7pt": 13, "mpi7pt": 13, "mpi13pt": 25, "mpi25pt": 49, "27pt": 53, "mpi125pt": 134}.get(name)
    if flop:
        tf = flop * cells / (ms * 1e-3) / 1e12
        r["roofline_fp64_vector"] = {"bound": "valu", "flop_per_point": flop, "achieved": tf, "peak": FP64_VECTOR_TFLOPS, "unit": "TFLOP/s",
                                     "frac": tf / FP64_VECTOR_TFLOPS}
    if O is not None:
        r["max_abs_err_vs_oracle"] = verify_sample(bl, O, sid, info, st, sto, step_e, off_out, ids, coeff, nsample=nsample)
    return r


def make_comm(args, world, dev):
    """(communicator of the C-ABI transport or None, transport name, fallback reason or None).
    rank 0's ncclUniqueId travels over the torch.distributed group; after that torch is not in the data path.  With
    --transport auto a failure to create it on ANY rank (librccl not loadable, ncclCommInitRank error) makes EVERY rank fall back
    to the torch transport: Comm.bootstrap_all_or_none cannot diverge before its all-reduce, so the decision is collective.
    One GPU with --self-via-comm: a 1-rank communicator; its torch fallback is a 1-rank nccl process group (main())."""
    via_self = world == 1 and args.self_via_comm
    if not via_self and (world == 1 or args.backend != "nccl"):
        return None, ("gloo-staged" if world > 1 else "device links"), None
    if args.transport == "torch":
        return None, "torch", None
    from bricklib_amd.decomp import Comm
    if args.test_fail_comm:          # test hook: the communicator cannot be created (what a missing librccl looks like)
        def _no_id():
            raise RuntimeError("--test-fail-comm: brk_comm_unique_id made to fail")
        Comm.unique_id = staticmethod(_no_id)
    if via_self:
        try:
            return Comm.init_rank(Comm.unique_id(), 0, 1), "abi: RCCL send/recv to self", None
        except Exception as e:
            comm, why = None, repr(e)
    else:
        comm, why = Comm.bootstrap_all_or_none(device=dev)
        if comm is not None:
            return comm, "abi", None
    if args.transport == "abi":
        raise RuntimeError(f"--transport abi: the C-ABI RCCL communicator could not be created on every rank ({why})")
    sys.stderr.write(f"bench.py: C-ABI communicator unavailable ({why}): all ranks use the torch transport\n")
    return None, "torch", why


def exchange_self_check(d, st, ex, rank, world, dev, label):
    """One exchange on recognisable data BEFORE anything is timed (reference rule, include/brick-mpi.h:476-491: ghost run i <- skin
    run i of the rank behind ghost[i].neighbor): every rank writes rank * nbricks + brick id into its surface bricks (all ranks have the same decomposition; exact in fp64
    far beyond any size that fits a GPU) and -1 into its
    ghost bricks, exchanges once and checks every ghost run against the ids its peer must have sent.  Returns None, or a text naming
    the first wrong run.  A mis-routed or mis-ordered run becomes a non-zero exit, not a silent drift."""
    s0, s1, s2 = d.sep_pos
    stride = int(s2)                       # ids of one rank live in [0, nbricks): rank * nbricks + id decodes unambiguously
    v = st.dat.view(-1, st.step)
    v[s0:s1] = (torch.arange(s0, s1, device=dev, dtype=torch.float64) + float(rank * stride))[:, None]
    v[s1:s2] = -1.0
    torch.cuda.synchronize()
    ex.exchange(torch.cuda.current_stream())
    torch.cuda.synchronize()
    want = np.empty(s2 - s1, dtype=np.float64)
    for g, k in zip(d.ghost, d.skin):
        src = d.rank_map[g["neighbor"]]
        want[g["pos"] - s1:g["pos"] - s1 + g["len"]] = np.arange(k["pos"], k["pos"] + g["len"], dtype=np.float64) + float(src * stride)
    bad = (v[s1:s2] != torch.from_numpy(want).to(dev)[:, None]).any(dim=1).cpu().numpy()
    if not bad.any():
        return None
    b = int(np.argmax(bad)) + s1
    for q, g in enumerate(d.ghost):
        if g["pos"] <= b < g["pos"] + g["len"]:
            got = float(v[b, 0].item())
            return (f"exchange self-check ({label}) FAILED on rank {rank}: ghost run {q} (neighbor set {g['neighbor']:#x}, from rank "
                    f"{d.rank_map[g['neighbor']]}), brick {b}: holds {got:.0f} = rank {int(got) // stride if got >= 0 else -1} brick {int(got) % stride if got >= 0 else -1}, "
                    f"expected rank {d.rank_map[g['neighbor']]} brick {d.skin[q]['pos'] + b - g['pos']}; {int(bad.sum())} ghost bricks wrong")
    return f"exchange self-check ({label}) FAILED on rank {rank}: brick {b} outside every ghost run"


def run_single(args, bl, dev):
    from bricklib_amd.brick import grid_to_device
    from bricklib_amd.stencils import STENCILS
    N = args.n
    name = args.stencil or "7pt"
    sid = STENCILS[name]
    if args.variant:
        from bricklib_amd._lib import check, lib
        if not hasattr(lib, "brk_set_variant"):
            raise SystemExit("--variant needs a development build of libbrk.so (make -C bricklib_amd/csrc VARIANTS=1)")
        check(lib.brk_set_variant(sid, args.variant))
    K = args.steps if args.steps is not None else 200
    W = args.warmup if args.warmup is not None else 20
    SB = (N + 16) // 8
    grid, info = bl.init_grid((SB, SB, SB), dev)
    if args.layout == "interleaved":
        st = sto = bl.BrickStorage.allocate(info.nbricks, 1024, dev)
        step_e, off_out = 1024, 512
    else:
        st, sto = bl.BrickStorage.allocate(info.nbricks, 512, dev), bl.BrickStorage.allocate(info.nbricks, 512, dev)
        step_e, off_out = 512, 0
    st.dat.uniform_(0.0, 1.0)
    bIn, bOut = bl.Brick(info, st, 0), bl.Brick(info, sto, off_out)
    from bricklib_amd.order import tiled
    # same bricks as the reference launch `b = grid[tk][tj][ti]` (3axis.hip.cpp:36-39); only the traversal order is ours
    ids = tiled(grid.reshape(SB, SB, SB)[1:-1, 1:-1, 1:-1])
    ids_dev = grid_to_device(ids, dev)
    rng = np.random.default_rng(1)
    coeff = rng.random(129) if name in ("7pt", "27pt") else None
    stream = torch.cuda.current_stream()

    NB = SB - 2
    box = (0, (SB, SB, SB), (1, 1, 1), (NB, NB, NB))     # the reference launch: dim3 block(NB,NB,NB), b = grid[GB+z][GB+y][GB+x]

    zz = bl.Zigzag()
    zigzag = bool(args.alternate) and not args.one_way

    def step(one_way=False, alternate=False):
        rev = (alternate or (zigzag and not one_way)) and zz.turn()
        if args.form == "box":
            bl.apply_stencil(sid, bIn, bOut, box=box, coeff=coeff, stream=stream, reverse=rev)
        else:
            bl.apply_stencil(sid, bIn, bOut, ids=ids_dev, coeff=coeff, stream=stream, reverse=rev)

    cells = float(N) ** 3
    extra = {}
    if args.steady_seconds > 0:
        # >= 2 s of back-to-back launches (clocks and memory at their sustained state).  Runs BEFORE the headline
        # windows: with the driver's --steps 20 a window is 8 ms, and measured straight after process start it sits on the clock
        # ramp; after this phase the W warm-up steps + K timed steps below see the device in the state a sweep loop runs in.
        for _ in range(max(W, 3)):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(20):
            step()
        e1.record(stream)
        torch.cuda.synchronize()
        est_ms = e0.elapsed_time(e1) / 20          # calibration only: how many launches make args.steady_seconds
        per = max(K, 100)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record(stream)
        n_l = 0
        while True:
            for _ in range(per):
                step()
            n_l += per
            if n_l * est_ms * 1e-3 >= args.steady_seconds:
                break
        e1.record(stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        extra["steady_state"] = {"seconds": dt, "launches": n_l, "kernel_ms": e0.elapsed_time(e1) / n_l,
                                 "value": cells * n_l / dt / 1e9, "unit": "GStencil/s",
                                 "roofline_frac": BYTES_PER_UPDATE * cells / (e0.elapsed_time(e1) / n_l * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # the reference's GPU protocol hiptime_func (stencils/stencils.hip.h:13-31): 1 warm-up + 100 launches between two events
        from bricklib_amd.timing import hiptime_func
        sec = hiptime_func(step, stream)
        extra["hiptime_func"] = {"seconds_per_launch": sec, "value": cells / sec / 1e9, "unit": "GStencil/s",
                                 "protocol": "hiptime_func (stencils/stencils.hip.h:13-31): warm-up + 100 launches between events"}
    for _ in range(W):
        step()
    torch.cuda.synchronize()
    # The timed region is exactly K steps between two synchronisations; it is repeated `--windows` times back to back and the
    # MEDIAN window is the headline (a single window of 20 launches is 8 ms: box noise moves it by a few %).
    # Every window is reported.
    wins = []
    for _ in range(max(1, args.windows)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(K):
            step()
        e1.record(stream)
        torch.cuda.synchronize()
        wins.append((time.perf_counter() - t0, e0.elapsed_time(e1) / K))
    order = sorted(range(len(wins)), key=lambda q: wins[q][0])
    wall, kern_ms = wins[order[len(order) // 2]]
    # the same launches in the other mode, beside the headline (not part of it): alternating directions when the headline is one
    # way (the default), one way when it alternates
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    side = (lambda: step(one_way=True)) if zigzag else (lambda: step(alternate=True))
    for _ in range(W):
        side()
    e0.record(stream)
    for _ in range(max(K, 50)):
        side()
    e1.record(stream)
    torch.cuda.synchronize()
    side_ms = e0.elapsed_time(e1) / max(K, 50)
    extra["one_way" if zigzag else "alternating"] = {
        "kernel_ms": side_ms, "value": cells / (side_ms * 1e-3) / 1e9, "unit": "GStencil/s",
        "roofline_frac": BYTES_PER_UPDATE * cells / (side_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "what": ("every launch walks the bricks in the same direction; the headline of this run alternates (--alternate)" if zigzag else
                 "successive sweeps in alternating directions (--alternate makes this the headline): a sweep starts in the part of the never-rewritten "
                 "input the Infinity Cache still holds, so this fraction is algorithmic bytes / time / 8 TB/s with part of the reads NOT served by HBM; "
                 "the headline (one direction) is the HBM number")}
    res = {"workload": f"single-{name}-{N}" + (f"-variant{args.variant}" if args.variant else ""), "layout": args.layout, "form": args.form,
           # what BrickStorage.allocate measured when it placed the output storage against the input storage (brk_storage_alloc, include/brk.h)
           "storage_placement": sto.placement, "sweep_direction": "alternating" if zigzag else "one way", "wall": wall, "K": K, "W": W, "cells_per_step": cells,
           "kernel_ms": kern_ms, "alg_bytes_per_launch": BYTES_PER_UPDATE * cells,
           # the name the launcher recorded for the last stencil launch of this thread (brk_last_kernel, ABI 1008): what actually ran
           "kernel": "%s over %d bricks" % (bl.last_kernel(), ids.size)}
    extra["timing"] = {"protocol": "%s%d warm-up steps, then the median of %d windows of %d launches, each between two device synchronisations"
                                   % ("the steady-state phase first; " if args.steady_seconds > 0 else "", W, len(wins), K),
                       "window_ms_per_step": [w / K * 1e3 for w, _ in wins], "window_kernel_ms": [k for _, k in wins]}
    O = None
    try:
        from oracle.oracle import Oracle
        O = Oracle()
        extra["max_abs_err_vs_oracle"] = verify_sample(bl, O, sid, info, st, sto, step_e, off_out, ids, coeff)
    except Exception as e:  # the checker is optional for the measurement itself
        extra["oracle_error"] = repr(e)
    if not args.no_config3 and args.form == "box":
        # BASELINE config 3 (27-pt + 125-pt at 512^3) and the other stencils of the path, on the same grid, outside the headline
        blk = {}
        for nm in ("27pt", "mpi125pt", "mpi13pt", "mpi25pt"):
            if nm != name:
                blk[nm] = side_stencil(bl, O, nm, info, st, sto, step_e, off_out, ids, box, stream, zigzag=zigzag)
        try:      # HBM-side traffic per launch of these kernels from the committed PMC passes (static, like roofline.traffic)
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_by_stencil.json")))
            for nm in blk:
                if N == 512 and nm in pt["traffic_bytes_per_launch"] and pt.get("sweep_direction") == blk[nm]["sweep_direction"]:
                    blk[nm]["roofline"]["traffic"] = pt["traffic_bytes_per_launch"][nm]
                    blk[nm]["roofline"]["traffic_source"] = "profiles/pmc_traffic_by_stencil.json"
        except Exception:
            pass
        extra["config3"] = {"workload": "single-{27pt,mpi125pt}-%d (+ 13/25-point), same box entry point" % N, **blk}
    if not args.no_scripts and args.form == "box":
        # stencil scripts held as DATA by the host, on the same grid, outside the headline: a script of the shape of the
        # reference's stencils/cond.py compiled at run time (hiprtc; star-shaped of radius 1 -> the headline kernel around the program)
        # beside the tuned d3cond kernel, and the interpreter.  Never fails the bench: a missing compiler is reported as such.
        try:
            extra["stencil_scripts"] = script_lines(bl, info, st, sto, step_e, off_out, ids, stream)
        except Exception as e:
            extra["stencil_scripts"] = {"error": repr(e)[:300]}
    if not args.no_cpu_baseline:
        # cpu_baseline = the reference's own CPU code (kind "reference"; built for 512^3 7-point sweeps where /root/reference was
        # present); cpu_baseline_port = the brick-layout restatement (kind "port") beside it.  Where the reference leg is not
        # available (another size or stencil) the port is the baseline of the line.
        ref_leg = None
        try:
            if name in ("7pt", "mpi7pt"):
                ref_leg = cpu_baseline_reference(N, args.cpu_seconds, name, coeff)
        except Exception as e:
            extra["cpu_baseline_reference_error"] = repr(e)
        port_leg = None
        if O is not None:
            try:
                port_leg = cpu_baseline(O, sid, info, st, step_e, ids, coeff, cells, min(args.cpu_seconds, 8.0) if ref_leg else args.cpu_seconds)
            except Exception as e:
                extra["oracle_error"] = repr(e)
        if ref_leg is not None:
            extra["cpu_baseline"] = ref_leg
            extra["cpu_baseline_port"] = port_leg
        else:
            extra["cpu_baseline"] = port_leg
    return res, extra


def rccl_library(comm):
    """which RCCL the C-ABI transport is bound to (brk_comm_library: an RCCL the process has already mapped -- torch's -- is
    preferred over loading a second instance); None when the run has no C-ABI communicator"""
    if comm is None:
        return None
    from bricklib_amd.decomp import Comm
    try:
        return Comm.library()
    except Exception as e:
        return repr(e)


def _phase_stats(rows, cycle_ms, world):
    """rows: per probed cycle a dict of phase -> ms on this rank; returns phase -> {min, avg, max, sigma} over ranks of the
    per-rank mean, the reference's statistics lines on stderr (rank 0), and the part of the exchange the timed cycles did not hide"""
    from bricklib_amd.timing import format_stats, rank_statistics
    keys = list(rows[0].keys())
    mine = {k: sum(r[k] for r in rows) / len(rows) for k in keys}
    out = {k: rank_statistics(mine[k], world) for k in keys}
    out["cycle_ms_timed"] = cycle_ms
    out["exposed_exchange_ms"] = max(0.0, cycle_ms - out["calc_ms"]["max"])
    out["what"] = ("per-cycle phase times measured AFTER the timed region, in order on one stream, HIP events (host_call_ms: host time "
                   "inside the transport call); {min, avg, max, sigma} over ranks of each rank's mean over %d cycles, the reference's "
                   "mpi_statistics (include/brick-mpi.h:758-793); exposed_exchange_ms = timed cycle - max calc" % len(rows))
    if int(os.environ.get("RANK", "0")) == 0 and (world > 1 or "WORLD_SIZE" not in os.environ or os.environ.get("WORLD_SIZE") == "1"):
        for k in keys:
            sys.stderr.write("bench.py: " + format_stats(k[:-3], out[k]) + "\n")
    return out


def phase_probe_weak(ex, calc, main, world, cycle_ms, cycles=5):
    import torch.distributed as dist
    from bricklib_amd.timing import marker
    rows = []
    ev = lambda: torch.cuda.Event(enable_timing=True)
    for it in range(cycles + 1):
        marks, host_call = [], 0.0
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        for st in range(ex.nstages):
            e = [ev() for _ in range(4)]
            e[0].record(main)
            with marker("pack"):
                ex.exchange_begin(main, st)
            e[1].record(main)
            t0 = time.perf_counter()
            with marker("transport"):
                ex.exchange_transport(main, st)
            host_call += time.perf_counter() - t0
            e[2].record(main)
            with marker("unpack"):
                ex.exchange_end(main, st)
            e[3].record(main)
            marks.append(e)
        c0, c1 = ev(), ev()
        c0.record(main)
        with marker("calc"):
            calc()
        c1.record(main)
        torch.cuda.synchronize()
        if it == 0:
            continue        # first probed cycle: warm-up
        rows.append({"pack_ms": sum(e[0].elapsed_time(e[1]) for e in marks), "transport_ms": sum(e[1].elapsed_time(e[2]) for e in marks),
                     "unpack_ms": sum(e[2].elapsed_time(e[3]) for e in marks), "calc_ms": c0.elapsed_time(c1), "host_call_ms": host_call * 1e3})
    return _phase_stats(rows, cycle_ms, world)


def phase_probe_strong(sr, transport, calc, main, world, cycle_ms, cycles=3):
    import torch.distributed as dist
    from bricklib_amd.timing import marker
    rows = []
    ev = lambda: torch.cuda.Event(enable_timing=True)
    for it in range(cycles + 1):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        e = [ev() for _ in range(5)]
        e[0].record(main)
        with marker("pack"):
            sr.exchange_begin(main)
        e[1].record(main)
        t0 = time.perf_counter()
        with marker("transport"):
            if sr.peers:
                transport(sr, main)
        host_call = time.perf_counter() - t0
        e[2].record(main)
        with marker("unpack"):
            sr.exchange_end(main)
        e[3].record(main)
        with marker("calc"):
            calc()
        e[4].record(main)
        torch.cuda.synchronize()
        if it:
            rows.append({"pack_ms": e[0].elapsed_time(e[1]), "transport_ms": e[1].elapsed_time(e[2]), "unpack_ms": e[2].elapsed_time(e[3]),
                         "calc_ms": e[3].elapsed_time(e[4]), "host_call_ms": host_call * 1e3})
    return _phase_stats(rows, cycle_ms, world)


def run_weak(args, bl, dev, rank, world):
    import torch.distributed as dist
    from bricklib_amd.decomp import Exchanger, rank_to_coo
    from bricklib_amd.drivers import weak_cycle
    from bricklib_amd.stencils import STENCILS
    N = args.n
    name = args.stencil or ("mpi7pt_x2" if args.fuse == 2 else "mpi7pt")
    sid = STENCILS[name]
    K = args.steps if args.steps is not None else 25
    W = args.warmup if args.warmup is not None else 3
    cart = bl.dims_create(world)
    d = bl.BrickDecomp((N, N, N), 8)
    bl.populate(d, cart, rank_to_coo(rank, cart))
    info = d.getBrickInfo(dev)
    stA, stB = info.allocate(512, dev), info.allocate(512, dev)
    A, B = bl.Brick(info, stA, 0), bl.Brick(info, stB, 0)
    comm, transport, fallback_reason = make_comm(args, world, dev)

    def make_exchanger(c):
        return Exchanger(d, stA, rank, mode=args.exchange, staged=args.backend == "gloo" and world > 1, comm=c,
                         self_via_comm=world == 1 and args.self_via_comm)

    ex = make_exchanger(comm)
    # ---- before anything is timed: which GPUs, and does one exchange put every peer's bricks where they belong? ---------------
    ranks_seen = None
    if world > 1:
        props = torch.cuda.get_device_properties(dev)
        me = {"rank": rank, "device": torch.cuda.current_device(), "name": props.name,
              "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
              "uuid": str(getattr(props, "uuid", ""))}
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)
        distinct = len({(x["pci"], x["uuid"]) for x in ranks_seen})
        informative = any(x["pci"] != "0000:00:00" or x["uuid"] not in ("", "None") for x in ranks_seen)   # a torch build that reports no ids proves nothing
        if args.backend == "nccl" and informative and distinct != world:      # every rank sees the same list: every rank leaves
            raise SystemExit(f"bench.py: {world} ranks on {distinct} distinct GPUs ({[x['pci'] for x in ranks_seen]}): one rank per GPU is required under RCCL")

    def checked_exchange(label):
        wd_ = _watchdog(min(args.watchdog_seconds, 240.0), rank, f"the exchange self-check ({label}): a send/recv that never completes")
        try:
            err = exchange_self_check(d, stA, ex, rank, world, dev, label)
        except Exception as e:
            err = f"exchange self-check ({label}) raised on rank {rank}: {e!r}"
        finally:
            wd_.cancel()
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return err, int(flag.item()) != 0

    err, any_bad = checked_exchange(transport)
    if any_bad and transport == "abi" and args.transport == "auto":
        # the C-ABI communicator exists but its exchange failed somewhere: every rank (the all-reduce above) moves to the torch transport
        if err:
            sys.stderr.write(err + "\n")
        fallback_reason = err or "the self-check failed on another rank"
        # the communicator whose exchange just failed may still have work stuck on the device: bound the wait for it
        wd_ = _watchdog(min(args.watchdog_seconds, 240.0), rank, "destroying the C-ABI communicator after a failed self-check")
        try:
            torch.cuda.synchronize()
            comm.destroy()
        finally:
            wd_.cancel()
        comm, transport = None, "torch"
        ex = make_exchanger(None)
        err, any_bad = checked_exchange(transport)
    if any_bad:
        if err:
            sys.stderr.write(err + "\n")
        raise SystemExit(4)
    self_check = {"passed": True, "transport": transport, "ghost_runs": len(d.ghost), "messages": ex.messages}
    stA.dat.uniform_(0.0, 1.0)
    if args.emulate_exchange_ms > 0:
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # calibrate torch.cuda._sleep
        torch.cuda._sleep(1000000)
        c0.record()
        torch.cuda._sleep(4000000)
        c1.record()
        torch.cuda.synchronize()
        spin = int(args.emulate_exchange_ms / (c0.elapsed_time(c1) / 4000000))
        real_exchange = ex.exchange

        def slow_exchange(stream=None):
            real_exchange(stream)
            with torch.cuda.stream(stream or torch.cuda.current_stream()):
                torch.cuda._sleep(spin)

        ex.exchange = slow_exchange
    from bricklib_amd.drivers import SweepLists, TrapezoidLists, weak_cycle_deep
    S = bl.stencil_iters(sid)
    sweep_ms = None

    def global_sum():
        """sum of the field over all ranks' subdomains (bricks [1, sep1)); the MPI_* star stencils have weights that add up
        to 1, so on the periodic global domain every sweep conserves it -- a size-independent check of the whole exchange +
        sweep schedule that also runs at 8 x 512^3"""
        t = stA.dat[512:512 * d.sep_pos[1]].sum().reshape(1)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    sum0 = global_sum()
    main = torch.cuda.current_stream()
    xstream = torch.cuda.Stream()            # the exchange runs here when a schedule overlaps it with sweeps
    schedule = "plain" if args.no_overlap else args.schedule
    deep_depth = args.deep_depth
    deep_early = args.deep_early == "1"      # "auto": priced below together with the schedule
    probe_ms = None
    if schedule == "auto":
        # measure, don't guess: the exchange alone (max over ranks) ...
        t0_, t1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(7):
            if it == 2:
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t0_.record(main)
            ex.exchange(main)
        t1_.record(main)
        torch.cuda.synchronize()
        # ... and one full sweep, to price the schedules: measured on one MI355X at 512^3 (profiles/r01_f_*), relative to
        # the in-order cycle T = S sweeps: surface costs +4.5 % and hides up to 0.21 T of exchange, deep costs +9 % and
        # hides up to 0.59 T; an overlapped exchange is taken to run 1.3x slower than alone (it shares HBM with the sweeps).
        from bricklib_amd.stencils import apply_stencil
        all_ids = SweepLists(d, dev).all
        s0_, s1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(5):
            if it == 2:
                s0_.record(main)
            apply_stencil(sid, A, B, ids=all_ids, stream=main)
        s1_.record(main)
        torch.cuda.synchronize()
        x = torch.tensor([t0_.elapsed_time(t1_) / 5, s0_.elapsed_time(s1_) / 3], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(x, op=dist.ReduceOp.MAX)
        probe_ms, sweep_ms = float(x[0].item()), float(x[1].item())
        T = S * sweep_ms
        tb = [d.t_dims[a] for a in range(3)]
        share = [max(0, (tb[0] - 2 * (t + 1))) * max(0, (tb[1] - 2 * (t + 1))) * max(0, (tb[2] - 2 * (t + 1))) / float(tb[0] * tb[1] * tb[2])
                 for t in range(1, S + 1)]                       # part of sweep t that can run before the ghosts land
        # while it overlaps with sweeps the exchange is taken to progress 1.2x slower (shared HBM): a window of h ms of
        # independent compute retires h / 1.2 ms of exchange
        # constants re-fitted in round 5 to the fused cycle at HEAD (profiles/r05_e_emulated_exchange_schedules.txt: cycle 1.84 ms; surface
        # costs +4.5 % and hides 0.35 T = 0.65 ms, deep costs +6 % at full depth and hides sum(share) x pass / 1.2 = 1.1 ms)
        cost = {("plain", 0): T + probe_ms, ("surface", 0): 1.045 * T + max(0.0, probe_ms - 0.35 * T)}
        for k in range(2, S + 1):                               # deep with k split passes: +6 % at full depth (measured)
            cost[("deep", k)] = (1.0 + 0.06 * k / S) * T + max(0.0, probe_ms - sum(share[:k]) * sweep_ms / 1.2)
        # deep with the early hand-over of the last sweep's surface (round 5, weak_cycle_deep(state=...)): the next exchange starts under
        # the rest of the last late pass: +0.2 ms hidden for +2.5 % of the cycle (profiles/r05_e_*: pays from ~1.3 ms of exchange on)
        cost[("deep+early", S)] = (1.0 + 0.06 + 0.025) * T + max(0.0, probe_ms - sum(share[:S]) * sweep_ms / 1.2 - 0.2)
        schedule, deep_depth = min(cost, key=cost.get)
        if schedule == "deep+early":
            schedule, deep_early = "deep", args.deep_early != "0"
    overlap = schedule != "plain"
    state = {}
    ic = 32 if name == "mpi25pt" else 0     # radius 4: rows of a block in i-chunks of 32 bricks (order.slabs: the k-halo planes stay in L2)
    kp = name in ("mpi13pt", "mpi25pt")     # the stars sweep k-paired lists (two bricks per wave, brk_stencil_apply_list_kpaired: 25-point -12 %, 13-point -1.9 %)
    if schedule == "deep":
        lists = TrapezoidLists(d, S, dev, kpair=kp, icols=ic)
        whole = SweepLists(d, dev, kpair=kp, icols=ic)
        deep_depth = S if not deep_depth else max(1, min(deep_depth, S))

        def step():
            return weak_cycle_deep(sid, d, A, B, ex.exchange, lists, main=main, comm=xstream, depth=deep_depth, full=whole, zigzag=args.weak_zigzag,
                                   state=state if deep_early else None)
    else:
        lists = SweepLists(d, dev, kpair=kp, icols=ic)

        def step():
            return weak_cycle(sid, d, A, B, ex.exchange, main=main, comm=xstream, overlap=overlap, lists=lists, state=state, zigzag=args.weak_zigzag)

    # sustained clocks before the contract's W warm-up cycles + K timed cycles: >= 0.25 s of plain sweeps A -> B (no communication;
    # A is not written, B is scratch between cycles).  With the driver's --steps 20 the timed region is 40 ms; straight after the
    # set-up it sat on the clock ramp (see run_single).
    from bricklib_amd.stencils import apply_stencil as _sweep
    _ids = SweepLists(d, dev).all
    _t = time.perf_counter()
    while time.perf_counter() - _t < 0.25:
        for _ in range(50):
            _sweep(sid, A, B, ids=_ids, stream=main)
        torch.cuda.synchronize()
    for _ in range(W):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(main)
    for _ in range(K):
        step()
    e1.record(main)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    # outside the timed region: where a cycle's time goes, phase by phase, in order on one stream -- the reference's pack / call /
    # wait / calc lines (weak/main.cpp:271-289) for a GPU-resident exchange: pack (links + packing launch), transport (the
    # ncclSend/ncclRecv group: host call time and device time), unpack, calc (the S passes with no exchange); min / avg / max /
    # sigma over ranks as mpi_statistics does.  The timed cycles above may overlap these phases (schedule); what they could
    # not hide is `exposed_exchange_ms` = timed cycle - calc.
    probe_lists = lists if isinstance(lists, SweepLists) else whole
    phases = phase_probe_weak(ex, lambda: weak_cycle(sid, d, A, B, lambda s_: None, main=main, comm=None, overlap=False, lists=probe_lists),
                              main, world, wall / K * 1e3)
    # ... and the exchange alone (device links + RCCL), for reading the scaling numbers
    xs, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    xs.record(main)
    for _ in range(5):
        ex.exchange(main)
    x1.record(main)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    exchange_ms = xs.elapsed_time(x1) / 5
    sum1 = global_sum()
    # weights of the MPI_* stencils add up to 1 (fake.h:11-33): 0.4+6(.1); .4+6(.07+.03); .1+6(.06+.045+.03+.015); C0..C9 x multiplicities
    drift = abs(sum1 - sum0) / abs(sum0) if name in ("mpi7pt", "mpi13pt", "mpi25pt", "mpi125pt", "mpi7pt_x2") else None
    upd = 2 if name.endswith("_x2") else 1                 # a fused pass updates every cell twice
    cells = float(N) ** 3 * S * upd * world
    # per-peer traffic of one exchange and the rate each peer link saw while the exchange ran alone (all peers at once)
    per_peer = {}
    for m in ex.plan.msgs:
        if m["is_send"]:
            per_peer[m["peer"]] = per_peer.get(m["peer"], 0) + m["bytes"]
    res = {"workload": f"weak-{name[:-3] if upd == 2 else name}-{N}", "storage_placement": stB.placement, "passes_per_step": S, "conservation_rel_drift": drift, "updates_per_pass": upd, "wall": wall, "K": K, "W": W, "cells_per_step": cells,
           "kernel_ms": e0.elapsed_time(e1) / K / S, "alg_bytes_per_launch": BYTES_PER_UPDATE * float(N) ** 3,   # per pass: one read + one write per cell
           "kernel": "%s passes over brick lists of BrickDecomp(%d^3): %d passes (%d sweeps) + 1 exchange per step; kernel_ms = step / %d" % (name, N, S, S * upd, S),
           "cart": cart, "overlap": overlap, "exchange": args.exchange, "transport": transport, "transport_fallback_reason": fallback_reason, "exchange_self_check": self_check,
           "schedule": schedule, "deep_depth": deep_depth if schedule == "deep" else None, "deep_early_handover": bool(deep_early) if schedule == "deep" else None, "schedule_probe_ms": probe_ms, "sweep_probe_ms": sweep_ms, "exchange_ms_alone": exchange_ms, "exchange_bytes_per_rank": sum(g["len"] for g in d.ghost) * 4096,
           "remote_bytes_per_rank": ex.bytes_remote, "messages_per_exchange": ex.messages, "phases": phases, "rccl_library": rccl_library(comm),
           "per_peer_send_bytes": {str(k): v for k, v in sorted(per_peer.items())},
           "per_peer_GBps_exchange_alone": {str(k): v / (exchange_ms * 1e-3) / 1e9 for k, v in sorted(per_peer.items())} if exchange_ms > 0 else None}
    extra = {}
    if world > 1:
        res["ranks_seen"] = ranks_seen            # gathered (and checked: N distinct GPUs under RCCL) before the timed region
        res["distinct_gpus"] = len({(x["pci"], x["uuid"]) for x in ranks_seen})
        if not args.no_scaling_base:
            # the like-for-like base of a scaling curve: the SAME workload (same stencil, fusion, sweeps per exchange) on one
            # GPU with a self-periodic 1x1x1 decomposition, measured here on rank 0's GPU outside the timed region
            base = [None]
            if rank == 0:
                import copy
                a1 = copy.copy(args)
                a1.steps, a1.warmup, a1.schedule, a1.emulate_exchange_ms = max(5, min(K, 10)), 2, "plain", 0.0
                try:
                    w1, _ = run_weak(a1, bl, dev, 0, 1)
                    base[0] = w1["cells_per_step"] * w1["K"] / w1["wall"] / 1e9
                except Exception as e:
                    base[0] = repr(e)
            dist.broadcast_object_list(base, src=0)
            if isinstance(base[0], float):
                value = cells * K / wall / 1e9
                extra["scaling_base"] = {"value": base[0], "unit": "GStencil/s", "workload": res["workload"] + " on ONE GPU (1x1x1 periodic, device links), rank 0's GPU",
                                         "updates_per_pass": upd}
                extra["scaling_efficiency"] = value / (world * base[0])
            else:
                extra["scaling_base"] = {"error": base[0]}
    if comm is not None:
        torch.cuda.synchronize()
        comm.destroy()
    return res, extra


def run_strong(args, bl, dev, rank, world):
    """strong/main.cu shape: global dom^3 in sub^3 subdomains (Z-Morton ranges per rank), one exchange + ST_ITER sweeps per
    step; all subdomains of a rank are swept by ONE launch per sweep (bricklib_amd/strong.py)."""
    import torch.distributed as dist
    from bricklib_amd.strong import StrongRank, nccl_transport, staged_transport
    from bricklib_amd.stencils import STENCILS
    name = args.stencil or "mpi13pt"
    sid = STENCILS[name]
    K = args.steps if args.steps is not None else 10
    W = args.warmup if args.warmup is not None else 2
    sr = StrongRank(args.strong_dom, args.strong_sub, rank, world, dev, alias_local=args.strong_local_ghosts == "aliased")
    sr.storages[0].dat.uniform_(0.0, 1.0)
    main = torch.cuda.current_stream()
    S = bl.stencil_iters(sid)

    comm = None
    if args.backend == "gloo":
        transport = staged_transport
    elif world > 1 and (cm := make_comm(args, world, dev))[0] is not None:
        from bricklib_amd.strong import rccl_transport
        comm = cm[0]
        transport = rccl_transport(comm)
    else:
        transport = nccl_transport

    def global_sum():   # see run_weak: the MPI_* stencils conserve the sum of the periodic global field
        v = sr.storages[0].dat.view(sr.nsub, sr.nb, 512)[:, 1:sr.d.sep_pos[1]].sum().reshape(1)
        if world > 1:
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
        return float(v.item())

    sum0 = global_sum()

    def step():
        sr.exchange(transport, main)
        sr.sweeps(sid, stream=main, zigzag=bool(args.alternate))

    for _ in range(W):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(main)
    for _ in range(K):
        step()
    e1.record(main)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    phases = phase_probe_strong(sr, transport, lambda: sr.sweeps(sid, stream=main, zigzag=bool(args.alternate)), main, world, wall / K * 1e3)
    sum1 = global_sum()
    drift = abs(sum1 - sum0) / abs(sum0) if name in ("mpi7pt", "mpi13pt", "mpi25pt", "mpi125pt", "mpi7pt_x2") else None
    upd = 2 if name.endswith("_x2") else 1
    cells = float(args.strong_dom) ** 3 * S * upd
    local_cells = float(args.strong_sub) ** 3 * sr.nsub
    res = {"workload": f"strong-{name}-{args.strong_dom}-{args.strong_sub}", "storage_placement": [st.placement for st in sr.storages], "passes_per_step": S, "wall": wall, "K": K, "W": W, "cells_per_step": cells,
           "kernel_ms": e0.elapsed_time(e1) / K / S, "alg_bytes_per_launch": BYTES_PER_UPDATE * local_cells,
           "kernel": "%s: %s sweeps over %d subdomains of %d^3 (+ghost) per rank, one launch per sweep; kernel_ms = step / %d"
                     % (bl.last_kernel(), name, sr.nsub, args.strong_sub, S),
           "sweep_lists": "k-paired (two bricks per wave)" if sr.swept_kpaired else "tiled",
           "scaling": "strong", "updates_per_pass": upd, "conservation_rel_drift": drift, "cells_per_gpu": int(local_cells), "subdomains_per_rank": sr.nsub, "remote_bytes_per_rank": sr.bytes_remote,
           # ghost runs between two subdomains of one rank: reached through the adjacency table ("aliased": neither copied nor swept)
           # or copied by device links as in the reference ("copied"); bricks per sweep over subdomains + ghost rings vs subdomain bricks
           "local_ghosts": args.strong_local_ghosts, "local_link_bytes_per_rank": sr.bytes_local,
           "swept_bricks_per_sweep": sr.swept_bricks, "subdomain_bricks": sr.nsub * (args.strong_sub // 8) ** 3, "phases": phases, "rccl_library": rccl_library(comm),
           "transport": ("abi" if comm is not None else ("gloo-staged" if args.backend == "gloo" else "torch")) if world > 1 else "device links"}
    if comm is not None:
        torch.cuda.synchronize()
        comm.destroy()
    return res, {}


def _watchdog(seconds, rank, what="the run"):
    """A rank that hangs (a collective that never completes, a peer that died) must not sit there forever: after `seconds`
    the process reports and exits with status 3.  os._exit from a timer thread: no re-exec, no retry, nothing touches the GPU."""
    import faulthandler
    import threading

    def fire():
        sys.stderr.write(f"bench.py: rank {rank} did not finish {what} within {seconds:.0f} s -- giving up (exit 3)\n")
        faulthandler.dump_traceback(file=sys.stderr)
        sys.stderr.flush()
        os._exit(3)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a CHILD process (torch.distributed.run,
    one rank per GPU -- the `mpirun -np N weak/cuda` of the reference, weak/main.cu:46-60,246-287), let it write to our stdout /
    stderr and leave with its return code.  This parent never touches the GPU (torch.cuda.device_count() does not initialise it
    on this image) and nothing is re-exec'd."""
    import subprocess
    if args.backend == "nccl":
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} over RCCL needs {args.gpus} visible GPUs, this host shows {have} "
                             f"(--backend gloo rehearses the multi-rank control flow on fewer)\n")
            raise SystemExit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")         # what torchrun would set anyway, without its warning
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                   # does not return
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks")
    import bricklib_amd as bl
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path)"
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()      # rehearsal: ranks share the GPUs that exist
    elif world > 1 and torch.cuda.device_count() < world // max(1, int(os.environ.get("NNODES", "1"))) and local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} (local {local}) has no GPU of its own: {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    wd = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        wd = _watchdog(args.watchdog_seconds, rank)
        tmo = datetime.timedelta(seconds=max(60.0, args.watchdog_seconds - 30.0))
        if args.backend == "gloo":
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
    elif args.self_via_comm and args.transport != "abi":
        # one GPU, self-neighbour runs through a communicator: the torch transport (and --transport auto's fallback to it) needs a
        # 1-rank RCCL process group
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    workload = args.workload or ("single" if world == 1 else "weak")
    if workload == "single":
        res, extra = run_single(args, bl, dev)
        if not args.no_weak_reference:
            # outside the timed region: the N > 1 workload on this one GPU (periodic self-exchange), so that the N > 1 lines of
            # a scaling run have a like-for-like base next to the single-sweep headline
            import copy
            a2 = copy.copy(args)
            a2.stencil, a2.steps, a2.warmup = None, 10, 2
            try:
                torch.cuda.empty_cache()
                w, _ = run_weak(a2, bl, dev, 0, 1)
                extra["weak_workload_on_this_gpu"] = {"value": w["cells_per_step"] * w["K"] / w["wall"] / 1e9, "unit": "GStencil/s",
                                                      "ms_per_step": w["wall"] / w["K"] * 1e3, "workload": w["workload"],
                                                      "updates_per_pass": w["updates_per_pass"], "schedule": w["schedule"],
                                                      "conservation_rel_drift": w["conservation_rel_drift"],
                                                      "exchange_self_check": w["exchange_self_check"]}
                if not args.no_weak_prediction:
                    # What the first multi-GPU run will be compared with: the same cycle with its exchange STRETCHED by a calibrated spin
                    # kernel on the exchange stream (one GPU cannot run an inter-GPU exchange), `--schedule auto` choosing as it will
                    # there.  predicted efficiency = cycle of this GPU alone (above) / stretched cycle: what a rank of an N-GPU weak run
                    # sustains if its exchange takes that long.  Which row applies is a property of the fabric: a 2x2x2 periodic rank
                    # sends 67 MB to each of its three face peers, 0.67 / 0.9 / 1.34 ms at 100 / 75 / 50 GB/s per link and direction.
                    rows = {}
                    for x_ms in (0.5, 1.0, 1.5, 2.0):
                        a3 = copy.copy(a2)
                        a3.schedule, a3.emulate_exchange_ms, a3.steps, a3.warmup = "auto", x_ms, 20, 3
                        torch.cuda.empty_cache()
                        w3, _ = run_weak(a3, bl, dev, 0, 1)
                        ms3 = w3["wall"] / w3["K"] * 1e3
                        rows["%.1f" % x_ms] = {"cycle_ms": ms3, "schedule": w3["schedule"], "deep_depth": w3.get("deep_depth"), "deep_early_handover": w3.get("deep_early_handover"),
                                               "predicted_efficiency": (w["wall"] / w["K"] * 1e3) / ms3}
                    extra["predicted_weak_efficiency"] = {
                        "by_emulated_exchange_ms": rows, "base_cycle_ms": w["wall"] / w["K"] * 1e3,
                        "what": "weak cycle of this one GPU (exchange + 8 sweeps as 4 fused passes, 512^3 + ghost per rank) with the exchange stretched by a spin "
                                "kernel on its stream, --schedule auto; efficiency = unstretched plain cycle / stretched cycle.  Not a scaling curve (none can be "
                                "measured on one GPU): the number the first 8-GPU run is to be compared with.  Face-peer message of a 2x2x2 periodic rank: 67 MB "
                                "= 0.67 / 0.9 / 1.34 ms at 100 / 75 / 50 GB/s per link and direction"}
            except Exception as e:
                extra["weak_workload_on_this_gpu"] = {"error": repr(e)}
    elif workload == "strong":
        res, extra = run_strong(args, bl, dev, rank, world)
    else:
        res, extra = run_weak(args, bl, dev, rank, world)
    if rank == 0:
        value = res["cells_per_step"] * res["K"] / res["wall"] / 1e9
        ach = res["alg_bytes_per_launch"] / (res["kernel_ms"] * 1e-3) / 1e9
        # the same from the host's clock around the K timed steps (what the driver can check against its own clock): algorithmic
        # bytes of all passes of a step / wall time per step
        ach_step = res["alg_bytes_per_launch"] * res.get("passes_per_step", 1) / (res["wall"] / res["K"]) / 1e9
        out = {
            "metric": "GStencil/s", "value": value, "unit": "GStencil/s", "n_gpus": world, "steps": res["K"],
            "warmup": res["W"], "ms_per_step": res["wall"] / res["K"] * 1e3, "higher_is_better": True,
            "scaling": res.get("scaling", "weak"), "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": res["workload"], "brick": "8x8x8", "cells_per_gpu": res.get("cells_per_gpu", args.n ** 3),
                       **{k: res[k] for k in ("layout", "form", "sweep_direction", "updates_per_pass", "cart", "overlap", "exchange", "schedule", "deep_depth", "deep_early_handover", "schedule_probe_ms", "sweep_probe_ms", "exchange_ms_alone", "exchange_bytes_per_rank",
                                             "remote_bytes_per_rank", "subdomains_per_rank", "conservation_rel_drift", "transport", "transport_fallback_reason", "exchange_self_check", "messages_per_exchange",
                                             "per_peer_send_bytes", "per_peer_GBps_exchange_alone", "distinct_gpus", "ranks_seen", "phases", "rccl_library",
                                             "local_ghosts", "local_link_bytes_per_rank", "swept_bricks_per_sweep", "subdomain_bricks", "storage_placement") if k in res}},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "frac_by_step": ach_step / HBM_PEAK_GBS,
                         "frac_of_measured_copy": ach / HBM_COPY_GBS, "traffic": None, "kernel": res["kernel"],
                         "kernel_ms": res["kernel_ms"], "algorithmic_bytes_per_launch": res["alg_bytes_per_launch"]},
        }
        # HBM-side traffic per launch from the committed PMC passes of this same command (tools/prof_pmc.sh); null when the
        # stored measurement is for another workload
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            # the stored measurement must be for THIS run's workload, entry form and sweep direction; otherwise null
            key = "%s/%s/%s" % (res["workload"], res.get("form", "-"), res.get("sweep_direction", "-"))
            if key in pt.get("runs", {}):
                out["roofline"]["traffic"] = pt["runs"][key]["traffic_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/pmc_traffic.json [%s]: %s" % (key, pt["runs"][key].get("source", ""))
        except Exception:
            pass
        out.update(extra)
        out.setdefault("cpu_baseline", None)
        print(json.dumps(out))
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    if wd is not None:
        wd.cancel()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # any failure (an RCCL error, a failed check) ends the rank with a non-zero status and a traceback; torch.distributed.run
        # then tears the other ranks down.  Nothing is retried and nothing is re-executed.
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
