"""Driver-level schedules of the reference, restated on device streams.

weak_cycle == one `brick_func` call of weak/main.cpp:172-213: exchange the ghosts of A,
then ST_ITER sweeps ping-ponging A/B, the last one skipping the outer brick ring, result back in A.  Thanks to
the region ordering of BrickDecomp the three brick sets involved are plain id ranges:
    [1, sep0)  inner bricks -- never read a ghost brick, so their first sweep overlaps the exchange
    [1, sep1)  the subdomain itself (what the last sweep computes; `skip = 1` in the reference)
    [1, sep2)  subdomain + ghost ring (what every other sweep computes)
"""
import torch

from .brick import grid_to_device
from .order import decomp_lists
from .stencils import apply_stencil, stencil_iters


class SweepLists:
    """Device-resident brick lists of one BrickDecomp for the sweep schedule (same brick sets as the id ranges in
    the module docstring, but in the cache-friendly order of bricklib_amd.order)."""

    def __init__(self, d, device, rows_per_block=8):
        for name, ids in decomp_lists(d, rows_per_block).items():
            setattr(self, name, grid_to_device(ids, device))


def weak_cycle(stencil, d, A, B, exchange, coeff=None, main=None, comm=None, overlap=True, lists=None, state=None):
    """A, B: Brick objects over two storages built from decomposition `d`; exchange(stream) refreshes A's ghosts.
    main/comm: torch streams (default: current stream, no overlap).  lists: SweepLists(d) (optional; without it the
    sweeps run over plain id ranges).  state: dict carried from one cycle to the next (optional): the last sweep writes
    the surface bricks first and records an event, so that the NEXT cycle's exchange can start while the inner bricks
    of the last sweep -- and then of the first sweep -- are still being computed (the exchange only reads surface bricks
    of A and writes ghost bricks of A, which the last sweep neither writes nor reads)."""
    S = stencil_iters(stencil)
    sep0, sep1, sep2 = d.sep_pos
    main = main or torch.cuda.current_stream()

    def sweep(src, dst, which):
        if lists is not None:
            apply_stencil(stencil, src, dst, ids=getattr(lists, which), coeff=coeff, stream=main)
        else:
            rng = {"inner": (1, sep0), "shell": (sep0, sep2), "surface": (sep0, sep1), "all": (1, sep2),
                   "domain": (1, sep1)}[which]
            apply_stencil(stencil, src, dst, brick_range=rng, coeff=coeff, stream=main)

    if comm is None or not overlap:
        exchange(main)
        sweep(A, B, "all")
    else:
        ev0 = state.pop("surface_done", None) if state is not None else None
        if ev0 is None:
            ev0 = torch.cuda.Event()
            ev0.record(main)             # A as left by the previous cycle
        comm.wait_event(ev0)
        exchange(comm)                   # ghosts of A arrive on the comm stream ...
        ev1 = torch.cuda.Event()
        ev1.record(comm)
        sweep(A, B, "inner")             # ... while the inner bricks sweep
        main.wait_event(ev1)
        sweep(A, B, "shell")
    for _ in range(S // 2 - 1):
        sweep(B, A, "all")
        sweep(A, B, "all")
    if state is not None and comm is not None and overlap:
        sweep(B, A, "surface")           # what the next exchange sends ...
        ev = torch.cuda.Event()
        ev.record(main)
        state["surface_done"] = ev
        sweep(B, A, "inner")             # ... is ready before the bulk of the last sweep
    else:
        sweep(B, A, "domain")
    return S
