"""The reference's three timing protocols, for drivers written against this package.

  time_func     stencils/stencils.h:40-53     CPU: warm-up, then batches of 1, 2, 4, ... calls until TOT_TIME has passed
  hiptime_func  stencils/stencils.hip.h:13-31 GPU: warm-up, HIP_ITER = 100 launches between two events
  time_mpi      stencils/fake.h:392-404       MPI: warm-up, MPI_ITER calls, seconds per call

The C++ twins live in include/bricktime.h."""
import time

TOT_TIME = 5.0     # stencils/stencils.h: #define TOT_TIME
HIP_ITER = 100     # stencils/stencils.hip.h:11
MPI_ITER_CPU, MPI_ITER_GPU = 25, 100   # weak/args.cpp defaults / weak/main.cu


def time_func(func, tot_time=TOT_TIME):
    """-> (seconds per call, calls made, elapsed).  The reference returns (ed - st) / (it - 1): with batches of 1, 2, 4, ...
    the calls made are exactly it - 1."""
    it = 1
    func()                      # warm up
    st = ed = time.perf_counter()
    calls = 0
    while ed < st + tot_time:
        for _ in range(it):
            func()
        calls += it
        it <<= 1
        ed = time.perf_counter()
    return (ed - st) / (it - 1), calls, ed - st


def hiptime_func(func, stream=None, iters=HIP_ITER):
    """seconds per launch: one warm-up call, device synchronise, `iters` calls between two events on `stream`
    (the reference records on the null stream; here the stream the launches go to)."""
    import torch
    stream = stream or torch.cuda.current_stream()
    func()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        func()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters / 1000.0


def time_mpi(func, iters, reset=None):
    """seconds per call of `func` (one exchange + ST_ITER sweeps): warm-up call, `reset()` (the reference zeroes its
    pack/call/wait/move/calc accumulators here), then `iters` calls.  Callers synchronise the device inside `func` or
    around this function, as the reference's brick_func does."""
    func()
    if reset is not None:
        reset()
    st = time.perf_counter()
    for _ in range(iters):
        func()
    return (time.perf_counter() - st) / iters


# ---- markers and per-phase statistics of the distributed drivers ------------------------------------------------------------------
class marker:
    """`with marker("pack"): ...` -- a roctx range around a phase of the exchange / sweep cycle, so that
    `rocprofv3 --marker-trace --kernel-trace -- python bench.py --gpus N ...` shows which kernels and RCCL calls belong to which
    phase.  PyTorch on ROCm routes torch.cuda.nvtx to roctx; without it the ranges cost nothing."""
    _push = _pop = None
    _ready = False

    def __init__(self, name):
        self.name = name
        if not marker._ready:
            marker._ready = True
            try:
                import torch
                torch.cuda.nvtx.range_push("brk")
                torch.cuda.nvtx.range_pop()
                marker._push, marker._pop = torch.cuda.nvtx.range_push, torch.cuda.nvtx.range_pop
            except Exception:
                marker._push = marker._pop = None

    def __enter__(self):
        if marker._push:
            marker._push(self.name)
        return self

    def __exit__(self, *exc):
        if marker._pop:
            marker._pop()
        return False


def rank_statistics(value, world=1, group=None):
    """min / avg / max / sigma of one number per rank, on every rank -- the reference's mpi_statistics (include/brick-mpi.h:758-793:
    sample standard deviation).  COLLECTIVE when world > 1: every rank of the torch.distributed group must call it (a one-rank
    run nested inside a larger job -- the scaling base of bench.py -- passes world = 1 and gathers nothing)."""
    import math
    vals = [float(value)]
    if world > 1:
        import torch.distributed as dist
        box = [None] * dist.get_world_size(group)
        dist.all_gather_object(box, float(value), group=group)
        vals = [float(v) for v in box]
    n = len(vals)
    avg = sum(vals) / n
    var = sum((v - avg) ** 2 for v in vals) / (n - 1) if n > 1 else 0.0
    return {"min": min(vals), "avg": avg, "max": max(vals), "sigma": math.sqrt(var)}


def format_stats(name, s, unit="ms"):
    """one line in the reference's format (weak/main.cpp:271-289: `calc [min, avg, max] (sigma: s)`)"""
    return "%s [%.6g, %.6g, %.6g] (σ: %.6g) %s" % (name, s["min"], s["avg"], s["max"], s["sigma"], unit)
