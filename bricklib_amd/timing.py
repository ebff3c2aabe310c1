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
